# A/B of bench.py argument sets on one box: REPS=3 bash scripts/ab_args.sh "--lw-kernel grouped" "--lw-kernel auto" ...   (interleaved)
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for a in "$@"; do
    i=$((i+1))
    python bench.py --steps 60 --warmup 5 --no-cpu-baseline $a > gpurun_out/aba_$i.json 2>gpurun_out/aba_$i.err
    python - "$a" $i <<'PY'
import json, sys
tag, i = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f"gpurun_out/aba_{i}.json"))
    k = d["roofline"]["kernel_ms"]
    print(f"{tag:32s} {d['ms_per_step']:.4f} ms/step | " + " ".join(f"{n.replace('zm_','').replace('modal_aero_','')}={v:.3f}" for n, v in k.items()), flush=True)
except Exception as e:
    print(tag, "FAILED", open(f"gpurun_out/aba_{i}.err").read()[-400:], flush=True)
PY
  done
done
