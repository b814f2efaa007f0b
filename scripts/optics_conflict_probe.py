#!/usr/bin/env python3
"""What do LDS bank conflicts cost modal_aero_sw?  

The SW optics gather four table corners x three tables per (band, mode) unit and lane; which corners a lane reads follows from
its cell's refractive index, i.e. from the cell's aerosol composition.  bench.py's synthetic field draws every cell's
composition independently (log-uniform over four decades per species), so the 16 lanes of a ds_read_b128 group read 16
unrelated 48-byte records: a record's three 16-byte pieces fall on 16-byte columns (48*i + {0,16,32}) mod 256, records i and j
collide iff i = j mod 16, and 16 independent draws over 16 classes have an expected maximum load of ~3 -- a birthday
problem that no fixed re-indexing of the records changes (any bijection of 70 records onto 16 classes keeps the statistics).

This probe runs the same kernel on the same grid with a COLUMN-COHERENT field (every level of a column has the composition of
its first level: the 32 / 64 lanes that share a column read the same corners, which the LDS broadcasts) and reports both
times; under `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE` the counters of both.
The time difference is everything a conflict-free layout could gain on the bench's field.

usage: python scripts/optics_conflict_probe.py [--field independent|coherent|both] [--ncol 48602] [--pver 32] [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from cam_nor_physics_amd import abi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--field", default="both", choices=["independent", "coherent", "both"])
    ap.add_argument("--ncol", type=int, default=48602)
    ap.add_argument("--pver", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch
    from cam_nor_physics_amd.engine import ColumnPhysics
    params = abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, synth.limcnv_from_pref(a.pver))
    eng = ColumnPhysics(params, 0)
    props = synth.make_aero_props(synth.DEFAULT_SEED)
    eng.aero_tables_set(props, synth.PHYSCONST)
    st = synth.make_state(a.ncol, a.pver, synth.DEFAULT_SEED)
    pdeldry = eng.dev(st["pdeldry"])
    nm = len(props["nspec"])
    for field in (("independent", "coherent") if a.field == "both" else (a.field,)):
        aer = synth.make_aero_state(a.ncol, a.pver, props, synth.DEFAULT_SEED)
        if field == "coherent":
            aer["specmmr"][:] = aer["specmmr"][..., 0:1]
            aer["qaerwat"][:] = aer["qaerwat"][..., 0:1]
        A = {k: eng.dev(aer[k]) for k in ("specmmr", "dgnumwet", "qaerwat")}
        out = {k: eng.empty(a.ncol, synth.NSWBANDS, a.pver + 1) for k in ("tauxar", "wa", "ga", "fa")}
        out["status"] = eng.empty(a.ncol, dtype=torch.int32)
        vis = eng.empty(3, nm, a.ncol, a.pver)
        for _ in range(3):
            eng.modal_aero_sw(pdeldry, A["specmmr"], A["dgnumwet"], A["qaerwat"], out=out, vis=vis)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            eng.modal_aero_sw(pdeldry, A["specmmr"], A["dgnumwet"], A["qaerwat"], out=out, vis=vis)
        e1.record()
        torch.cuda.synchronize()
        print(f"modal_aero_sw field={field} ncol={a.ncol} pver={a.pver}: {e0.elapsed_time(e1) / a.reps:.4f} ms per launch "
              f"(tauxar checksum {float(out['tauxar'].sum()):.6e})", flush=True)


if __name__ == "__main__":
    main()
