#!/usr/bin/env python3
"""Timeline of the host face at ne30 size for a few slice counts (COLPHYS_HF_TRACE=1 prints per-slice phase times)."""
import os, sys, json
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")]
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from cam_nor_physics_amd import abi, synth
from cam_nor_physics_amd.engine import ColumnPhysics
from cam_nor_physics_amd.hostface import HF_AERO, HF_SWDIAG, HF_ZM, HostFace, chunk_counts
pver, pcols, ncol = 32, 16, 48602
eng = ColumnPhysics(abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, synth.limcnv_from_pref(pver)), 0)
props = synth.make_aero_props(synth.DEFAULT_SEED)
st = synth.make_state(ncol, pver, synth.DEFAULT_SEED, moist_boost=1.1)
aer = synth.make_aero_state(ncol, pver, props, synth.DEFAULT_SEED)
eng.aero_tables_set(props, synth.PHYSCONST)
what = HF_ZM | HF_AERO | HF_SWDIAG if "--zm" not in sys.argv else HF_ZM
for ns in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16".split(","))]:
    hf = HostFace(eng, chunk_counts(ncol, pcols), pcols, pver, what, ns)
    hf.fill_from_synth(st, aer, props)
    hf.run(synth.ZTODT)
    os.environ["COLPHYS_HF_TRACE"] = "0"
    ts = [hf.run(synth.ZTODT) for _ in range(4)]
    os.environ["COLPHYS_HF_TRACE"] = "1"
    t = hf.run(synth.ZTODT)
    os.environ["COLPHYS_HF_TRACE"] = "0"
    print(ns, "slices: wall", [round(x["wall_ms"], 2) for x in ts], "cpu", round(t["host_cpu_ms"], 3), "h2d", round(t["h2d_ms"], 2), "phys", round(t["physics_ms"], 2),
          "d2h", round(t["d2h_ms"], 2), "GB", round(t["h2d_bytes"] / 1e9, 3), round(t["d2h_bytes"] / 1e9, 3), flush=True)
    hf.close()
