#!/bin/bash
# bench lines of the BASELINE configs the default command does not run (one GPU each), for profiles/<tag>_configs.jsonl
# usage (GPU box): bash scripts/config_lines.sh <tag>
OUT=gpurun_out/configs_$1; mkdir -p $OUT; : > $OUT/lines.jsonl
run() { echo "# $1" >> $OUT/lines.jsonl; shift; python3 bench.py --no-cpu-baseline --no-dropin "$@" >> $OUT/lines.jsonl 2>> $OUT/err.txt || exit 1; }
run "configs[1]: 48602 x L32, ZM only" --no-aero && \
run "configs[2]: ne120, 777602 x L32, geopotential_t + ZM" --ncol 777602 --no-aero --steps 10 --warmup 2 && \
run "ne120 with the optics" --ncol 777602 --steps 5 --warmup 1 && \
run "configs[4]: 48602 x L70" --pver 70 --steps 20 --warmup 3 && \
run "configs[3] per-GPU share of the 8-way static split: 6075 x L32" --ncol 6075 && \
run "4-way share: 12151 x L32" --ncol 12151 && \
run "2-way share: 24301 x L32" --ncol 24301 && \
run "rounds 1-2 field (10 % deep)" --moist-boost 1.0
python3 - $OUT/lines.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"):
        print(l.strip()); continue
    d = json.loads(l)
    print("   value %.4g %s  ms_per_step %.4f  deep %.3f  roofline.frac %.4f traffic %s" % (d["value"], d["unit"], d["ms_per_step"], d["config"]["deep_fraction"], d["roofline"]["frac"], d["roofline"]["traffic"]))
PY
