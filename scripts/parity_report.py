#!/usr/bin/env python3
"""Prints the worst relative error (tests/util.py metric) of the HIP optics and zm_convr against the committed
reference goldens, for the library in COLPHYS_LIB (A/B of experiment builds)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from util import rel_err_profile
from cam_nor_physics_amd import abi, synth
from cam_nor_physics_amd.engine import ColumnPhysics

for name in cases.AERO_GOLDEN:
    g = cases.load_golden(name)
    pver = g["in_t"].shape[1]
    eng = ColumnPhysics(cases.default_params(pver), 0)
    eng.aero_tables_set(cases.aero_props_from_golden(g), synth.PHYSCONST)
    d = eng.dev
    a = [d(g["in_" + k]) for k in ("pdeldry", "specmmr", "dgnumwet", "qaerwat")]
    sw = {k: v.cpu().numpy() for k, v in eng.modal_aero_sw(*a).items()}
    lw = {k: v.cpu().numpy() for k, v in eng.modal_aero_lw(*a).items()}
    errs = {k: rel_err_profile(sw[k], g["sw_" + k]) for k in ("tauxar", "wa", "ga", "fa")}
    errs["lw"] = rel_err_profile(lw["tauxar"], g["lw_tauxar"])
    print(name, {k: f"{v:.2e}" for k, v in errs.items()})

# ---- zm_convr against the reference goldens, and against the C oracle on a large fresh sample ---------------
import test_parity_gpu as T
for name in cases.ZM_GOLDEN:
    g = cases.load_golden(name)
    n, pver = g["in_t"].shape
    eng = T.engine(cases.golden_params(g))
    o = T.run_zm(eng, cases.zm_args_from_golden(g))
    ref = {k[3:]: v for k, v in g.items() if k.startswith("zm_")}
    print(name, "zm_convr worst rel err vs reference golden: %.2e" % T.check_zm(o, ref, n))

sys.path.insert(0, os.path.join(ROOT, "oracle"))
from c_oracle import COracle
ncol = int(os.environ.get("PARITY_NCOL", "20000"))
for pver, seed in ((32, 101), (70, 102)):
    params = cases.default_params(pver)
    eng = T.engine(params)
    co = COracle(params)
    st = synth.make_state(ncol if pver == 32 else ncol // 4, pver, seed)
    ga = cases.geo_args(st)
    zi_r, zm_r = co.geopotential_t(*ga, st["q"], st["rair"], synth.PHYSCONST["gravit"], st["zvir"], dycore=abi.DYCORE_FV)
    args = cases.zm_args_from_state(st, zi_r, zm_r)
    ref = co.zm_convr(*args)
    o = T.run_zm(eng, args)
    same = np.array_equal(o["lengath"], ref["lengath"])
    denom = np.maximum(np.abs(ref["cape"]), 1e-3 * np.abs(ref["cape"]).max())
    e = np.abs(o["cape"] - ref["cape"]) / denom
    eh = np.abs(o["heat"] - ref["heat"]).max(axis=1) / np.maximum(np.abs(ref["heat"]).max(axis=1), 1e-3 * np.abs(ref["heat"]).max())
    print(f"L{pver}: {st['t'].shape[0]} columns, {int(ref['lengath'].sum())} deep, gathered sets identical: {same}; "
          f"cape rel err max {e.max():.2e}, columns > 1e-12: {(e > 1e-12).sum()}, > 1e-11: {(e > 1e-11).sum()}, > 1e-10: {(e > 1e-10).sum()}; "
          f"heat rel err max {eh.max():.2e}, > 1e-11: {(eh > 1e-11).sum()}, > 1e-10: {(eh > 1e-10).sum()}")
