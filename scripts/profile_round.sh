#!/bin/bash
# One round's judged artefacts, written under gpurun_out/prof_<tag>/ (copy the summaries into profiles/):
#   kernel_stats (default command), kernel_stats_serial (--serial --no-graph: un-overlapped per-kernel averages, the ones
#   bench.py's HIP-event numbers must agree with), PMC passes, and the bench line itself.
# usage (on the GPU box): bash scripts/profile_round.sh <tag>
export TMPDIR=/tmp
TAG=$1; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dropin > $OUT/bench_under_rocprof.json 2> $OUT/kt.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dropin --serial --no-graph > $OUT/bench_serial_under_rocprof.json 2> $OUT/kts.err && \
bash scripts/pmc_passes.sh prof_$TAG/pmc --serial --no-graph > $OUT/pmc.log 2>&1 && \
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.csv && \
python3 scripts/pmc_stamp.py $OUT/pmc_summary.csv
find $OUT -name "*kernel_stats.csv"
