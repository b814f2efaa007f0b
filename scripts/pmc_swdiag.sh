# PMC passes over scripts/bench_swdiag.py (modal_aero_sw and its diagnostics kernel); summaries under gpurun_out/
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_swd_$i -- python3 scripts/bench_swdiag.py 32 > gpurun_out/pmc_swd_$i.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/pmc_swd_$i | grep -E "kernel|modal_aero_sw"
done
