"""Times modal_aero_sw and its history-diagnostics kernel at the ne30 size (48 602 columns) with HIP events.
Usage: python scripts/bench_swdiag.py [pver]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402

from cam_nor_physics_amd import synth  # noqa: E402
from cam_nor_physics_amd.engine import ColumnPhysics  # noqa: E402

pver = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ncol = 48602
eng = ColumnPhysics(cases.default_params(pver), 0)
props = synth.make_aero_props(3)
st = synth.make_state(ncol, pver, 3)
aer = synth.make_aero_state(ncol, pver, props, 3)
eng.aero_tables_set(props, synth.PHYSCONST)
d = eng.dev
a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
pm, t, tl = d(st["pmid"]), d(st["t"]), d(aer["troplevchem"].astype(np.int32))
vis = eng.empty(3, eng.tables.nmodes, ncol, pver)
sw = eng.modal_aero_sw(*a, vis=vis)
out = eng.modal_aero_sw_diag(a[0], pm, t, a[1], a[2], a[3], tl, sw["tauxar"], vis=vis)
torch.cuda.synchronize()
# durations from the library's own HIP events around each launch (colphys_profile_*), 20 launches each
eng.profile_enable(True)
for name, key, fn in (("modal_aero_sw", "modal_aero_sw", lambda: eng.modal_aero_sw(*a, out=sw)),
                      ("modal_aero_sw + hand-over", "modal_aero_sw", lambda: eng.modal_aero_sw(*a, out=sw, vis=vis)),
                      ("modal_aero_sw_diag (tables)", "modal_aero_sw_diag",
                       lambda: eng.modal_aero_sw_diag(a[0], pm, t, a[1], a[2], a[3], tl, sw["tauxar"], out=out)),
                      ("modal_aero_sw_diag (hand-over)", "modal_aero_sw_diag",
                       lambda: eng.modal_aero_sw_diag(a[0], pm, t, a[1], a[2], a[3], tl, sw["tauxar"], out=out, vis=vis))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    eng.profile_reset()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name:32s} L{pver}: {eng.profile()[key][0]:.4f} ms")
