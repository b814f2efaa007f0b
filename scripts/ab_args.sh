# A/B of bench arguments on one box: bash scripts/ab_args.sh "<args1>" "<args2>" ...   (each run twice, interleaved)
for rep in 1 2; do
  i=0
  for a in "$@"; do
    i=$((i+1))
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline $a > gpurun_out/ab_$i.json 2>gpurun_out/ab_$i.err
    python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/ab_$i.json')); print(repr('$a'), d['ms_per_step'])
except Exception as e:
    print(repr('$a'), 'FAILED', open('gpurun_out/ab_$i.err').read()[-300:])"
  done
done
