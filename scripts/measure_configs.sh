mkdir -p gpurun_out/meas
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/meas/def_$i.json 2>/dev/null; done
python bench.py --no-cpu-baseline --no-aero --steps 40 --warmup 5 > gpurun_out/meas/noaero.json 2>/dev/null
python bench.py --no-cpu-baseline --ncol 777602 --no-aero --steps 10 --warmup 2 > gpurun_out/meas/ne120_noaero.json 2>/dev/null
python bench.py --no-cpu-baseline --ncol 777602 --steps 10 --warmup 2 > gpurun_out/meas/ne120.json 2>/dev/null
python bench.py --no-cpu-baseline --pver 70 --steps 20 --warmup 3 > gpurun_out/meas/l70.json 2>/dev/null
python bench.py --no-cpu-baseline --lw-kernel auto --steps 40 --warmup 5 > gpurun_out/meas/lwauto.json 2>/dev/null
python scripts/bench_swdiag.py 32 > gpurun_out/meas/swdiag32.txt 2>/dev/null
python scripts/bench_swdiag.py 70 > gpurun_out/meas/swdiag70.txt 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/meas/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/meas/*.txt")): print(open(f).read())
PY
