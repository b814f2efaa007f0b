#!/bin/bash
# the probe's two fields, each under one PMC pass; summary -> gpurun_out/$1/summary.txt   usage: bash scripts/optics_conflict_pmc.sh <dir>
export TMPDIR=/tmp
OUT=gpurun_out/${1:-optics_conflict}
mkdir -p $OUT
python3 scripts/optics_conflict_probe.py --field both > $OUT/time.txt 2>$OUT/time.err || exit 1
for f in independent coherent; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
      --output-format csv -d $OUT/$f -- python3 scripts/optics_conflict_probe.py --field $f --reps 3 > $OUT/$f.log 2>$OUT/$f.err || exit 1
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
print(open(out + "/time.txt").read().strip())
for f in ("independent", "coherent"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for p in glob.glob(f"{out}/{f}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "modal_aero_reg_kernel" in r["Kernel_Name"] and "ILb1E" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    row = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
    print(f, "per launch:", {k: f"{v:.4g}" for k, v in sorted(row.items())})
    if row.get("SQ_ACTIVE_INST_LDS"):
        print(f, "bank-conflict cycles / LDS-instruction-active cycles =", round(row["SQ_LDS_BANK_CONFLICT"] / row["SQ_ACTIVE_INST_LDS"], 3))
PY
