"""TEST / MEASUREMENT INFRASTRUCTURE -- the CPU side of bench.py's `cpu_baseline` leg.

Runs the same physics substep as bench.py's GPU step on a slice of the same synthetic columns, with
  kind = "reference": oracle/_ref (the reference's own Fortran, chunked at pcols=16 exactly as CAM's chunk
                      loop feeds it, physpkg.F90:1147-1161), or
  kind = "port":      the C restatement (oracle/colphys_oracle.c),
one worker process per host core (the reference parallelises the chunk loop with OpenMP threads; separate
processes are used here because the accessor stubs behind modal_aer_opt keep per-chunk module state).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for _p in (_ROOT, _HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

_G = {}


def _substep(o, st, aer, lo, hi, c, dt, with_aero):
    """o: RefOracle or COracle (same call surface).  Mirrors bench.py:substep."""
    s = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == st["ncol"] else v) for k, v in st.items()}
    ga = (s["lnpint"], s["lnpmid"], s["pint"], s["pmid"], s["pdel"], s["rpdel"])
    z = o.zm_convr(s["t"], s["q"], s["pblh"], s["zm"], s["phis"], s["zi"], s["pmid"], s["pint"], s["pdel"], 0.5 * dt,
                   s["tpert"], s["landfrac"])
    t1 = s["t"] + z["heat"] * dt / c["cpair"]
    q1 = np.maximum(s["q"] + z["qtnd"] * dt, 1.0e-12)
    o.geopotential_t(*ga, t1, q1, s["rair"], c["gravit"], s["zvir"])
    e = o.zm_conv_evap(t1, s["pmid"], s["pdel"], q1, s["landfrac"], z["rprd"], s["cld"], dt, z["prec"])
    t1 = t1 + e["tend_s"] * dt / c["cpair"]
    q1 = np.maximum(q1 + e["tend_q"] * dt, 1.0e-12)
    o.geopotential_t(*ga, t1, q1, s["rair"], c["gravit"], s["zvir"])
    m = o.momtran(s["u"], s["v"], z["mu"], z["md"], z["du"], z["eu"], z["ed"], z["dp"], z["dsubcld"], z["jt"], z["maxg"],
                  z["ideep"], z["lengath"], dt)
    t1 = t1 + m["seten"] * dt / c["cpair"]
    zi, zm = o.geopotential_t(*ga, t1, q1, s["rair"], c["gravit"], s["zvir"])
    chk = float(z["heat"].sum() + e["tend_q"].sum() + m["seten"].sum() + zm.sum())
    if with_aero:
        a = {k: (v[..., lo:hi, :] if k != "troplevchem" else v[lo:hi]) for k, v in aer.items()}
        sw = o.modal_aero_sw(s, a)                      # the reference's modal_aero_sw computes its history diagnostics inside
        if not hasattr(o, "lib") or not hasattr(o.lib, "ref_dims"):      # the C restatement keeps them in a second function
            chk += float(o.modal_aero_sw_diag(s, a)["col"].sum())
        lw = o.modal_aero_lw(s, a)
        chk += float(sw["tauxar"].sum() + lw["tauxar"].sum())
    return chk, int(z["lengath"].sum())


def _init(kind, pver, with_aero):
    from cam_nor_physics_amd import abi, synth
    nl, c = synth.ZM_NAMELIST, synth.PHYSCONST
    limcnv = synth.limcnv_from_pref(pver)
    if kind == "reference":
        from ref_oracle import RefOracle
        o = RefOracle(16, pver)
        o.zm_convi(limcnv, nl)
    else:
        from c_oracle import COracle
        o = COracle(abi.make_zm_params(c, nl, limcnv))
    if with_aero:
        o.aero_tables_set(_G["props"], c)
    _G["o"] = o


def _work(rng):
    from cam_nor_physics_amd import synth
    lo, hi = rng
    return _substep(_G["o"], _G["st"], _G["aer"], lo, hi, synth.PHYSCONST, synth.ZTODT, _G["with_aero"])


def run_cpu_baseline(st, aer, props, pver, with_aero=True, cores=None, sample_cols=None, kind=None):
    """Returns dict(value=columns/s, cores, kind, sample, seconds).  Must be called BEFORE the process
    touches the GPU (workers are forked).  kind: "reference" (oracle/_ref), "port" (the C restatement), or None = the
    reference build where it is present, else the port."""
    import multiprocessing as mp
    from ref_oracle import have_ref
    if kind is None:
        kind = "reference" if have_ref(16, pver) else "port"
    if kind == "reference" and not have_ref(16, pver):
        return None
    ncol = st["ncol"] if sample_cols is None else min(sample_cols, st["ncol"])
    if not cores:                       # every core this process may use (bench.py passes min(affinity, cgroup quota))
        try:
            cores = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores = os.cpu_count() or 1
    _G.update(st=st, aer=aer, props=props, with_aero=with_aero)
    # slices aligned to the reference's chunk width
    nchunks = (ncol + 15) // 16
    per = (nchunks + cores - 1) // cores
    ranges = [(i * per * 16, min((i + 1) * per * 16, ncol)) for i in range(cores) if i * per * 16 < ncol]
    ctx = mp.get_context("fork")
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)            # the reference prints its zm_convi banner on the log unit (stderr)
    try:
        with ctx.Pool(len(ranges), initializer=_init, initargs=(kind, pver, with_aero)) as pool:
            pool.map(_work, [(r[0], min(r[0] + 16, r[1])) for r in ranges])     # warm-up: one chunk each
            t0 = time.perf_counter()
            res = pool.map(_work, ranges)
            dt = time.perf_counter() - t0
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)
    return dict(value=ncol / dt, unit="columns/s", cores=len(ranges), kind=kind,
                sample=f"{ncol} columns x L{pver} (the whole bench workload of one rank), one substep, "
                       f"{len(ranges)} processes, chunks of 16",
                seconds=dt, deep=sum(r[1] for r in res))
