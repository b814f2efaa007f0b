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
    eng.aero_tables_set(synth.make_aero_props(int(g["seed"])), synth.PHYSCONST)
    d = eng.dev
    a = [d(g["in_" + k]) for k in ("pdeldry", "specmmr", "dgnumwet", "qaerwat")]
    sw = {k: v.cpu().numpy() for k, v in eng.modal_aero_sw(*a).items()}
    lw = {k: v.cpu().numpy() for k, v in eng.modal_aero_lw(*a).items()}
    errs = {k: rel_err_profile(sw[k], g["sw_" + k]) for k in ("tauxar", "wa", "ga", "fa")}
    errs["lw"] = rel_err_profile(lw["tauxar"], g["lw_tauxar"])
    print(name, {k: f"{v:.2e}" for k, v in errs.items()})
