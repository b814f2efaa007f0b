#!/usr/bin/env python3
"""Basic-block census of one kernel in a hipcc -S listing: per block the VALU / SALU / LDS / VMEM / branch / waitcnt
counts and the back edges (loops).  Usage: isa_blocks.py file.s kernel_substring [min_valu]"""
import re
import sys

path, name = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*:", l) and name in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], {"label": "entry", "ins": [], "line": start}
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "ins": [], "line": i}
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*)", l)
    if m and not l.strip().startswith(";") and not l.strip().startswith("."):
        cur["ins"].append((m.group(1), m.group(2)))
blocks.append(cur)
idx = {b["label"]: n for n, b in enumerate(blocks)}
tot = {}
for n, b in enumerate(blocks):
    c = {"valu": 0, "f64": 0, "trans": 0, "salu": 0, "lds": 0, "vmem": 0, "br": 0, "wait": 0}
    tgt = []
    for op, args in b["ins"]:
        if op.startswith("v_"):
            c["valu"] += 1
            if "f64" in op:
                c["f64"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp|rndne|cvt|trig|exp|log|fract|floor|ceil)", op):
                c["trans"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            c["br"] += 1
            t = args.split()[-1] if args else ""
            tgt.append(t)
        elif op.startswith("s_waitcnt"):
            c["wait"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    b["c"], b["tgt"] = c, tgt
    for k, v in c.items():
        tot[k] = tot.get(k, 0) + v
print("kernel", name, "blocks", len(blocks), "totals", tot)
for n, b in enumerate(blocks):
    back = [t for t in b["tgt"] if t in idx and idx[t] <= n]
    if b["c"]["valu"] >= minv or back:
        print(f"{n:4d} {b['label']:12s} L{b['line']:6d} {b['c']}  -> {b['tgt']} {'BACK ' + str(back) if back else ''}")
