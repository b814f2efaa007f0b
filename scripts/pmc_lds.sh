export TMPDIR=/tmp
for v in ov_base abl_notable abl_nodma; do
  COLPHYS_LIB=cam-nor-physics_amd/_build/$v.so rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/pmc_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial --no-graph > gpurun_out/pmc_$v.log 2>&1
  echo "== $v"; python3 scripts/pmc_summary.py gpurun_out/pmc_$v | grep -E "kernel|modal_aero_sw"
done
