#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-layout (Fortran-facing) entry points: ONE colphys_*_chunks call over all 3 038 chunks
of a 48 602-column rank, arrays in the reference's (pcols, pver, nchunks) layout on the host.  The time includes
packing, the H2D copies, the kernels, the D2H copies and the scatter back (never `value`)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from cam_nor_physics_amd import abi, synth
from cam_nor_physics_amd.engine import ColumnPhysics

ncol, pver, pcols = 48602, 32, 16
nch = (ncol + pcols - 1) // pcols
ncols = np.full(nch, pcols, dtype=np.int32); ncols[-1] = ncol - pcols * (nch - 1)
eng = ColumnPhysics(cases.default_params(pver), 0)
st = synth.make_state(ncol, pver, 20261004)
args = cases.zm_args_from_state(st, st["zi"], st["zm"])


def to_chunks(a, nlev):
    out = np.full((nch, nlev, pcols), np.inf)
    flat = np.asarray(a, dtype=np.float64).reshape(ncol, nlev)
    pad = np.full((nch * pcols, nlev), np.inf); pad[:ncol] = flat
    out[:] = pad.reshape(nch, pcols, nlev).transpose(0, 2, 1)
    return np.ascontiguousarray(out)


vals = dict(t=args[0], qh=args[1], pblh=args[2], zm=args[3], geos=args[4], zi=args[5], pap=args[6], paph=args[7], dpp=args[8],
            tpert=args[10], landfrac=args[11])
hin = {k: to_chunks(vals[k], pver) for k in ("t", "qh", "pap", "dpp", "zm")}
hin.update({k: to_chunks(vals[k], pver + 1) for k in ("paph", "zi")})
hin.update({k: to_chunks(vals[k], 1) for k in ("geos", "pblh", "tpert", "landfrac")})
hout = {k: np.zeros((nch, pver, pcols)) for k in abi.ZM_CONVR_OUT_MID}
hout.update({k: np.zeros((nch, pver + 1, pcols)) for k in abi.ZM_CONVR_OUT_INT})
hout.update({k: np.zeros((nch, 1, pcols)) for k in abi.ZM_CONVR_OUT_COL})
hout.update({k: np.zeros((nch, 1, pcols), dtype=np.int32) for k in abi.ZM_CONVR_OUT_ICOL})
hout["lengath"] = np.zeros(nch, dtype=np.int32)
io = abi.ZmConvrIO()
for k, v in {**hin, **hout}.items():
    setattr(io, k, abi.iptr(v) if v.dtype == np.int32 else abi.dptr(v))


def call():
    rc = eng.lib.colphys_zm_convr_chunks(C.c_int(nch), C.c_int(pcols), C.c_int(pver), abi.iptr(ncols), C.c_double(args[9]), C.byref(io))
    assert rc == 0, eng.lib.colphys_last_error()


call()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    call()
dt = (time.perf_counter() - t0) / n
print(f"colphys_zm_convr_chunks, {nch} chunks x {pcols} (48 602 columns x L32), host layout, PCIe-inclusive: "
      f"{dt * 1e3:.1f} ms per call = {ncol / dt / 1e6:.2f} M columns/s; deep columns {int(hout['lengath'].sum())}")

# the legacy per-chunk entry: ONE chunk per call (what a chunk loop that calls the drop-in module per chunk does)
io1 = abi.ZmConvrIO()
one = {k: np.ascontiguousarray(v[:1]) for k, v in {**hin, **hout}.items() if k != "lengath"}
one["lengath"] = np.zeros(1, dtype=np.int32)
for k, v in one.items():
    setattr(io1, k, abi.iptr(v) if v.dtype == np.int32 else abi.dptr(v))
n1 = np.array([pcols], dtype=np.int32)
for _ in range(5):
    assert eng.lib.colphys_zm_convr_chunks(C.c_int(1), C.c_int(pcols), C.c_int(pver), abi.iptr(n1), C.c_double(args[9]), C.byref(io1)) == 0
t0 = time.perf_counter()
n = 200
for _ in range(n):
    eng.lib.colphys_zm_convr_chunks(C.c_int(1), C.c_int(pcols), C.c_int(pver), abi.iptr(n1), C.c_double(args[9]), C.byref(io1))
dt1 = (time.perf_counter() - t0) / n
print(f"colphys_zm_convr_chunks, ONE chunk of {pcols} columns per call: {dt1 * 1e6:.0f} us per call = {pcols / dt1 / 1e3:.1f} k columns/s "
      f"({nch} such calls = {nch * dt1 * 1e3:.0f} ms per substep of a rank)")
