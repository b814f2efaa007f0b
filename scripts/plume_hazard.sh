#!/bin/bash
# Round 2's "plume LDS hazard", reproduced on purpose and explained (cam_nor_physics_amd/isa_check.py).
# Builds variants of the plume kernels -- {register carry, LDS read-back of the qd / sd recurrence} x {wave_sync, __syncthreads}
# x serial-loop unroll factors x an extra fence inside the recurrence loop -- and for each compares zm_convr's outputs with two
# columns per wave (COLPHYS_PLUME_CPW=2) against one column per wave and against the shipped library.  Printed next to each
# variant: what the ISA scanner finds in it.  The two columns agree exactly when the scanner finds nothing: the differences are a
# backend miscompile (a live-range-split copy ahead of a loop-exit block's EXEC restore), not an LDS ordering problem, and the
# kind of barrier plays no part.
# NOTE: the variants miscompile at commit e8b916b (where profiles/r03_plume_hazard.txt was taken).  The later re-packing of the
# plume kernels' LDS profiles changed the register allocation and none of them trips the backend any more (the scanner reports
# 0 hits for all of them, and all agree); to reproduce, take cam-nor-physics_amd/csrc/k_zm_convr.hip from that commit.
# usage (GPU box): [HAZARD_VARIANTS="reg_wave lds_wave ..."] bash scripts/plume_hazard.sh
set -e
mkdir -p gpurun_out/hazard && rm -f gpurun_out/hazard/result.txt /tmp/hazard_base.npz
python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from cam_nor_physics_amd.build import build_hip, BUILD
import os
import sys
VARIANTS = {"reg_wave": [], "lds_wave": ["-DPLUME_QDSD_LDS"], "reg_barrier": ["-DPLUME_SYNC_BARRIER"],
            "lds_barrier": ["-DPLUME_QDSD_LDS", "-DPLUME_SYNC_BARRIER"],
            "lds_unroll1": ["-DPLUME_QDSD_LDS", "-DSERIAL_UNROLL=1"], "lds_syncloop": ["-DPLUME_QDSD_LDS", "-DPLUME_QDSD_SYNC"],
            "reg_unroll1": ["-DSERIAL_UNROLL=1"], "reg_unroll2": ["-DSERIAL_UNROLL=2"], "reg_unroll8": ["-DSERIAL_UNROLL=8"],
            "lds_syncloop_unroll1": ["-DPLUME_QDSD_LDS", "-DPLUME_QDSD_SYNC", "-DSERIAL_UNROLL=1"]}
want = os.environ.get("HAZARD_VARIANTS", "reg_wave reg_barrier reg_unroll1 reg_unroll2 reg_unroll8 lds_wave lds_barrier lds_unroll1 lds_syncloop").split()
from cam_nor_physics_amd import isa_check
with open("gpurun_out/hazard/isa.txt", "w") as f:
    for tag, fl in ((t, VARIANTS[t]) for t in want):
        build_hip(extra_flags=fl, out=os.path.join(BUILD, f"libcolphys_hz_{tag}.so"))
        hits = isa_check.check_sources(fl, sources=["k_zm_convr.hip"])
        f.write(f"{tag} {' '.join(fl) or '(shipped flags)'}: isa_check {len(hits)} hit(s)" +
                "".join(f"\n    {k.split('EEEv')[0][-22:]}: {ins}" for _, k, _, ins in hits) + "\n")
PY
for tag in shipped ${HAZARD_VARIANTS:-reg_wave reg_barrier reg_unroll1 reg_unroll2 reg_unroll8 lds_wave lds_barrier lds_unroll1 lds_syncloop}; do
  lib=$PWD/cam-nor-physics_amd/_build/libcolphys_hz_$tag.so
  [ $tag = shipped ] && lib=$PWD/cam-nor-physics_amd/_build/libcolphys_hip.so
  COLPHYS_LIB=$lib python3 - "$tag" <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/hazard/result.txt
import os, sys
sys.path[:0] = [".", "tests", "oracle"]
import numpy as np
import cases
from cam_nor_physics_amd import synth
from test_parity_gpu import engine, run_zm
tag = sys.argv[1]
pver, ncol = 32, 48602
eng = engine(cases.default_params(pver))
st = synth.make_state(ncol, pver, 77)
args = cases.zm_args_from_state(st, st["zi"], st["zm"])
os.environ["COLPHYS_PLUME_CPW"] = "1"; a = run_zm(eng, args)
os.environ["COLPHYS_PLUME_CPW"] = "2"; b = run_zm(eng, args)
def cmp(x, y):
    out = {}
    for k in x:
        if not np.array_equal(x[k], y[k]):
            d = x[k] != y[k]
            rel = np.abs(x[k][d].astype(float) - y[k][d]) / np.maximum(np.abs(x[k][d].astype(float)), 1e-300)
            out[k] = (int(d.sum()), float(rel.max()), tuple(int(v) for v in np.argwhere(d)[0]))
    return out
base = "/tmp/hazard_base.npz"
if tag == "shipped":
    np.savez(base, **a)
    ref = a
else:
    ref = dict(np.load(base)) if os.path.exists(base) else a
bad = cmp(a, b)
print(tag, "two columns per wave vs one:", "IDENTICAL" if not bad else f"DIFFER (count, max rel, first index) {bad}")
for nm, r in (("cpw1", a), ("cpw2", b)):
    bad = cmp(ref, r)
    print(tag, nm, "vs shipped library cpw1:", "IDENTICAL" if not bad else f"DIFFER {bad}")
PY
done
cat gpurun_out/hazard/isa.txt | tee -a gpurun_out/hazard/result.txt
