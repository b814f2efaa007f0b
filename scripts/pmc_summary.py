#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (scripts/pmc_passes.sh) per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections, re
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        if not any(k in name for k in ("zm_", "geopot", "momtran", "modal_aero", "physics_update")):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(acc)
ctrs = sorted({c for n in names for c in acc[n]})
w = csv.writer(sys.stdout, lineterminator="\n")      # kernel names carry commas (template arguments): quoted
w.writerow(["kernel"] + ctrs)
for n in names:
    w.writerow([n] + [f"{sum(acc[n][c])/len(acc[n][c]):.4g}" if acc[n][c] else "" for c in ctrs])
