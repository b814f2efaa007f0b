# A/B of environment settings on one box: [BENCH_ARGS=...] bash scripts/ab_env.sh "VAR=1 VAR2=2" "VAR=2" ...   (interleaved, REPS rounds)
ARGS="$BENCH_ARGS"
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS > gpurun_out/abe_$i.json 2>gpurun_out/abe_$i.err
    python - "$e" $i <<'PY'
import json, sys
tag, i = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f"gpurun_out/abe_{i}.json"))
    k = d["roofline"]["kernel_ms"]
    print(f"{tag:44s} {d['ms_per_step']:.4f} ms/step | " + " ".join(f"{n}={k[n]:.3f}" for n in ("zm_trigger", "zm_ascent2", "zm_plume2", "modal_aero_sw", "modal_aero_lw") if n in k), flush=True)
except Exception as e:
    print(tag, "FAILED", open(f"gpurun_out/abe_{i}.err").read()[-400:], flush=True)
PY
  done
done
