#!/bin/bash
# rocprofv3 PMC passes over the bench command (counters only, each in its own run; MI355X_MICROARCH.md "rocprofv3 PMC slots")
# usage: bash scripts/pmc_passes.sh <outdir-under-gpurun_out> [bench args]
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-dropin $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/p1.log 2>$OUT/p1.err && \
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/p2.log 2>$OUT/p2.err && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 bench.py $ARGS > $OUT/p3.log 2>$OUT/p3.err && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 bench.py $ARGS > $OUT/p4.log 2>$OUT/p4.err
find $OUT -name "*counter_collection.csv" | head
