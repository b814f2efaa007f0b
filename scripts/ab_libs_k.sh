# like ab_libs.sh, printing every kernel's un-overlapped time: bash scripts/ab_libs_k.sh lib1.so lib2.so ...
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    COLPHYS_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_$tag.json 2>gpurun_out/ab_$tag.err
    python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ab_{tag}.json"))
    k = d["roofline"]["kernel_ms"]
    print(f"{tag:16s} {d['ms_per_step']:.4f} | " + " ".join(f"{n.replace('zm_','').replace('modal_aero_','')}={v:.3f}" for n, v in k.items()), flush=True)
except Exception as e:
    print(tag, "FAILED", open(f"gpurun_out/ab_{tag}.err").read()[-400:], flush=True)
PY
  done
done
