#!/usr/bin/env python3
"""Times zm_conv_tend_2 (convective transport of N tracers on resident data) behind one zm_conv_tend on
the bench's synthetic ne30-shaped state: python scripts/bench_convtran.py [ncnst ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from cam_nor_physics_amd import abi, synth
    from cam_nor_physics_amd.engine import ColumnPhysics
    ncol, pver = int(os.environ.get("NCOL", 48602)), int(os.environ.get("PVER", 32))
    st = synth.make_state(ncol, pver, synth.DEFAULT_SEED)
    eng = ColumnPhysics(abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, synth.limcnv_from_pref(pver)), 0)
    d = eng.dev
    S = {k: d(st[k]) for k in ("t", "q", "u", "v", "pint", "pmid", "pdel", "rpdel", "lnpint", "lnpmid", "pdeldry", "phis",
                               "pblh", "tpert", "landfrac", "cld", "rair", "zvir", "zi", "zm")}
    S["ps"] = d(st["pint"][:, -1].copy())
    z = eng.zm_conv_tend(S, synth.ZTODT, pcols=16)
    ndeep = int(z["lengath"].sum().item())
    sizes = [int(a) for a in sys.argv[1:]] or [3, 31, 41]
    nmax = max(sizes)                                   # one allocation for every size (views): no allocator traffic between cases
    gen = torch.Generator(device=eng.device).manual_seed(7)
    qall = torch.rand((nmax, ncol, pver), dtype=torch.float64, device=eng.device, generator=gen) * 1e-6 + 1e-9
    frall = torch.rand((nmax, ncol, pver), dtype=torch.float64, device=eng.device, generator=gen)
    outall = torch.empty_like(qall)
    for ncnst in sizes:
        q, fr, out = qall[:ncnst], frall[:ncnst], outall[:ncnst]
        lq = np.ones(ncnst, dtype=np.int32); lq[0] = 0
        dry = (np.arange(ncnst) % 3 == 0).astype(np.int32)
        args = (q, fr, S["pdeldry"], z["mu"], z["md"], z["du"], z["eu"], z["ed"], z["dp"], z["jt"], z["maxg"], z["ideep"], lq, dry, synth.ZTODT)
        for _ in range(3):
            eng.zm_conv_tend_2(*args, ptend_q=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            eng.zm_conv_tend_2(*args, ptend_q=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        byts = (ncnst - 1) * (ndeep * pver * 8 * 3 + ncol * pver * 8)       # q, fracis in + dqdt out on deep columns, zero fill elsewhere
        print(f"zm_conv_tend_2: {ncnst - 1:3d} tracers, {ncol} columns ({ndeep} deep) x L{pver}: {ms:.3f} ms/call, "
              f"{ncol / ms * 1e3 / 1e6:.1f} M columns/s, {byts / ms / 1e6:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
