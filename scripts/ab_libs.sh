# A/B of library builds on one box: [BENCH_ARGS="--serial --no-graph"] bash scripts/ab_libs.sh lib1.so lib2.so ...   (interleaved, REPS rounds each)
# Build the variants first: python -c "from cam_nor_physics_amd.build import build_hip; build_hip(extra_flags=[...], out='.../lib_x.so')"
ARGS="$BENCH_ARGS"
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    COLPHYS_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS > gpurun_out/ab_$tag.json 2>gpurun_out/ab_$tag.err
    python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ab_{tag}.json"))
    k = d["roofline"]["kernel_ms"]
    print(f"{tag:28s} {d['ms_per_step']:.4f} ms/step | " + " ".join(f"{n}={k[n]:.3f}" for n in ("zm_trigger", "zm_ascent2", "zm_plume2", "zm_conv_evap", "modal_aero_sw", "modal_aero_lw") if n in k), flush=True)
except Exception as e:
    print(tag, "FAILED", open(f"gpurun_out/ab_{tag}.err").read()[-400:], flush=True)
PY
  done
done
