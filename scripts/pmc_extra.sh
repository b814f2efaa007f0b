#!/bin/bash
# extra PMC passes (instruction cache, scalar path, thread-level VALU use): bash scripts/pmc_extra.sh <outdir> [bench args]
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/x1 -- python3 bench.py $ARGS > $OUT/x1.log 2>$OUT/x1.err && \
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQC_DCACHE_MISSES --output-format csv -d $OUT/x2 -- python3 bench.py $ARGS > $OUT/x2.log 2>$OUT/x2.err && \
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/x3 -- python3 bench.py $ARGS > $OUT/x3.log 2>$OUT/x3.err
