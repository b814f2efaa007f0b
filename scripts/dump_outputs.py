#!/usr/bin/env python3
"""Runs the bench substep once on the synthetic ne30-shaped input and dumps every output to an .npz, so that two
library builds (COLPHYS_LIB) or two settings can be compared bit for bit:
    COLPHYS_LIB=a.so python scripts/dump_outputs.py gpurun_out/a.npz; COLPHYS_LIB=b.so python scripts/dump_outputs.py gpurun_out/b.npz
    python scripts/dump_outputs.py --compare gpurun_out/a.npz gpurun_out/b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    worst = 0.0
    nbit = 0
    for k in A.files:
        x, y = A[k], B[k]
        same = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
        if same:
            continue
        nbit += 1
        if x.dtype.kind == "f":
            d = np.abs(x - y)
            scale = np.maximum(np.abs(x).max(axis=tuple(range(1, x.ndim)), keepdims=True), 1e-300) if x.ndim > 1 else np.maximum(np.abs(x), 1e-300)
            rel = float(np.nanmax(d / scale))
            worst = max(worst, rel)
            print(f"  {k:12s} differs in {int((x != y).sum())} of {x.size} elements, max rel (per column profile) {rel:.3e}")
        else:
            print(f"  {k:12s} INTEGER MISMATCH in {int((x != y).sum())} elements")
            worst = np.inf
    print(f"{len(A.files)} arrays, {nbit} not bit-identical, worst relative difference {worst:.3e}")
    return 0 if worst < 1e-10 else 1


def main():
    if sys.argv[1] == "--compare":
        raise SystemExit(compare(sys.argv[2], sys.argv[3]))
    import torch
    from cam_nor_physics_amd import abi, synth
    from cam_nor_physics_amd.engine import ColumnPhysics
    ncol = int(os.environ.get("NCOL", 48602))
    pver = int(os.environ.get("PVER", 32))
    st = synth.make_state(ncol, pver, synth.DEFAULT_SEED)
    params = abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, synth.limcnv_from_pref(pver))
    eng = ColumnPhysics(params, 0)
    d = eng.dev
    S = {k: d(st[k]) for k in ("t", "q", "u", "v", "pint", "pmid", "pdel", "rpdel", "lnpint", "lnpmid", "pdeldry", "phis",
                               "pblh", "tpert", "landfrac", "cld", "rair", "zvir", "zi", "zm")}
    S["ps"] = d(st["pint"][:, -1].copy())
    out = eng.zm_conv_tend(S, synth.ZTODT, pcols=16)
    props = synth.make_aero_props(synth.DEFAULT_SEED)
    aer = synth.make_aero_state(ncol, pver, props, synth.DEFAULT_SEED)
    eng.aero_tables_set(props, synth.PHYSCONST)
    A = {k: d(aer[k]) for k in ("specmmr", "dgnumwet", "qaerwat")}
    sw = eng.modal_aero_sw(S["pdeldry"], A["specmmr"], A["dgnumwet"], A["qaerwat"])
    lw = eng.modal_aero_lw(S["pdeldry"], A["specmmr"], A["dgnumwet"], A["qaerwat"])
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res.update({"sw_" + k: v.cpu().numpy() for k, v in sw.items()})
    res.update({"lw_" + k: v.cpu().numpy() for k, v in lw.items()})
    np.savez(sys.argv[1], **res)
    print("wrote", sys.argv[1], "deep columns:", int(res["lengath"].sum()))


if __name__ == "__main__":
    main()
