#!/usr/bin/env python3
"""Overlap timeline of the replayed substep from a rocprofv3 --kernel-trace CSV (bench.py default command):
mean start / end / duration of every kernel of one graph replay, microseconds from the trigger's start.
usage: timeline.py <dir containing *_kernel_trace.csv> [skip_first_n_replays]"""
import csv, glob, re, sys, collections
root = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)
# a replay starts at each zm_trigger dispatch
steps, cur = [], None
for r in rows:
    n = short(r["Kernel_Name"])
    if n.startswith("zm_trigger"):
        cur = []
        steps.append(cur)
    if cur is not None and any(k in n for k in ("zm_", "geopot", "momtran", "modal_aero", "physics_update", "convtran")):
        cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
# bench.py: one eager substep (workspace allocation), then warmup + steps graph replays, then eager passes: keep the
# graph replays only (dispatches 1 .. nreplay), minus the first `skip`
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 12
# the graph replays: the steps whose kernels ran on more than one queue (ZM chain || optics); a capture without the second
# stream has none, then the most frequent kernel multiset (eager warm-up and the per-kernel timing passes launch other sets)
multi = [s for s in steps if len({q for *_, q in s}) > 1]
pool = multi or steps
shape = collections.Counter(tuple(sorted(n for n, *_ in s)) for s in pool).most_common(1)[0][0]
steps = [s for s in pool if tuple(sorted(n for n, *_ in s)) == shape][skip:nrep]
def keyed(step):
    seen, out = collections.Counter(), {}
    for n, a, b, q in step:
        out[(n, seen[n])] = (a - step[0][1], b - step[0][1], q)
        seen[n] += 1
    return out
ks = [keyed(s) for s in steps]
print(f"# {len(steps)} graph replays of {len(steps[0])} kernels; microseconds from the trigger's start (mean)")
print(f"{'kernel':34s}{'q':>3s}{'start':>10s}{'end':>10s}{'dur':>10s}")
order = sorted(ks[0], key=lambda k: sum(x[k][0] for x in ks))
for k in order:
    st = sum(x[k][0] for x in ks) / len(ks) / 1e3
    en = sum(x[k][1] for x in ks) / len(ks) / 1e3
    print(f"{k[0][:33]:34s}{ks[0][k][2]:>3s}{st:10.1f}{en:10.1f}{en - st:10.1f}")
