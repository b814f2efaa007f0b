#!/usr/bin/env python3
"""Randomised geometries of the host face against the resident path (the checks of tests/test_hostface_gpu.py): chunk widths 1..64,
ragged chunks anywhere, 4..128 levels, any slice count, every flag combination.  usage: python scripts/hostface_fuzz.py [n] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import cases
import test_hostface_gpu as T
from cam_nor_physics_amd import synth
from cam_nor_physics_amd.hostface import HF_AERO, HF_HIST, HF_SWDIAG, HF_ZM, HostFace

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for it in range(n):
    pver = int(rng.choice([4, 9, 26, 32, 33, 47, 70, 128]))
    pcols = int(rng.choice([1, 2, 3, 8, 16, 17, 32, 33, 48, 64]))
    nch = int(rng.integers(1, 40))
    ncols = rng.integers(1, pcols + 1, nch).astype(np.int32)
    if rng.uniform() < 0.5:
        ncols[: max(1, nch - 2)] = pcols
    what = int(rng.choice([HF_ZM, HF_ZM | HF_HIST, HF_ZM | HF_AERO, HF_ZM | HF_AERO | HF_SWDIAG | HF_HIST, HF_AERO, HF_AERO | HF_SWDIAG]))
    nslices = int(rng.integers(0, min(nch, 9) + 1))
    nmodes = int(rng.choice([1, 4, 4, 6]))
    ncol = int(ncols.sum())
    eng = T.engine(cases.default_params(pver))
    with_zm, with_aero, sw_diag, hist = bool(what & HF_ZM), bool(what & HF_AERO), bool(what & HF_SWDIAG), bool(what & HF_HIST)
    props = synth.make_aero_props(3 + it, nspec_m=tuple(int(x) for x in rng.integers(1, 7, nmodes)))
    st = synth.make_state(ncol, pver, 100 + it, moist_boost=1.15)
    aer = synth.make_aero_state(ncol, pver, props, 3 + it) if with_aero else None
    if with_aero:
        eng.aero_tables_set(props, synth.PHYSCONST)
    r = T.resident(eng, st, aer, props, pcols, sw_diag, hist)
    hf = HostFace(eng, ncols, pcols, pver, what, nslices)
    try:
        hf.fill_from_synth(st, aer, props)
        hf.run(synth.ZTODT)
        same = bool((ncols[:-1] == pcols).all())
        T.compare(hf, r, ncols, pcols, pver, with_aero, sw_diag, hist, same, with_zm)
    finally:
        hf.close()
    print(f"ok {it}: L{pver} pcols {pcols} chunks {nch} cols {ncol} what {what} slices {nslices} modes {nmodes}", flush=True)
print("hostface_fuzz: all agree")
