"""The Fortran side of the whole-node dispatch (cam-nor-physics_amd/fortran/colphys_resident.F90): a Fortran caller uploads
the physics_state fields of ALL chunks of a rank from CAM's (pcols,pver,nchunks) layout once, makes ONE
colphys_zm_conv_tend_dev call, and downloads ptend_all + the surface / pbuf fields (tests/fortran/resident_driver.F90).
Must agree bit for bit with the same engine driven from Python on the dense layout, ragged last chunk included."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
from cam_nor_physics_amd import abi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PCOLS = 16


def to_cam(a, nchunks, pad=np.inf):
    """[ncol][nlev] (or [ncol]) -> C-ordered (nchunks, nlev, pcols) = Fortran (pcols, nlev, nchunks), padding = +inf as in CAM"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    ncol, nlev = a.shape
    out = np.full((nchunks, nlev, PCOLS), pad)
    for c in range(nchunks):
        n = min(PCOLS, ncol - c * PCOLS)
        out[c, :, :n] = a[c * PCOLS:c * PCOLS + n].T
    return out


def from_cam(a, ncol):
    nchunks, nlev, _ = a.shape
    out = np.empty((ncol, nlev))
    for c in range(nchunks):
        n = min(PCOLS, ncol - c * PCOLS)
        out[c * PCOLS:c * PCOLS + n] = a[c, :, :n].T
    return out


def call_res_zm_conv_tend(lib, st, ncol, pver):
    """tests/fortran/resident_driver.F90 res_zm_conv_tend on CAM-layout copies of a dense state"""
    nchunks = (ncol + PCOLS - 1) // PCOLS
    ncols = np.array([min(PCOLS, ncol - c * PCOLS) for c in range(nchunks)], dtype=np.int32)
    names_in = ("t", "q", "u", "v", "pmid", "pint", "pdel", "rpdel", "lnpint", "lnpmid", "zm", "zi", "phis", "ps", "rair", "zvir",
                "pblh", "tpert", "landfrac", "cld")
    host = dict(st, ps=st["pint"][:, -1].copy())
    ins = [to_cam(host[k], nchunks) for k in names_in]
    names_out = ("ptend_s", "ptend_q", "ptend_u", "ptend_v", "prec", "snow", "rliq", "jctop", "jcbot", "mconzm", "evapcdp", "mu")
    nlev = dict(ptend_s=pver, ptend_q=pver, ptend_u=pver, ptend_v=pver, prec=1, snow=1, rliq=1, jctop=1, jcbot=1, mconzm=pver + 1,
                evapcdp=pver, mu=pver)
    outs = {k: np.full((nchunks, nlev[k], PCOLS), np.nan) for k in names_out}
    jt = np.zeros((nchunks, PCOLS), dtype=np.int32); maxg = np.zeros_like(jt); ideep = np.zeros_like(jt)
    lengath = np.zeros(nchunks, dtype=np.int32)
    nbad = C.c_int(-7)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    lib.res_zm_conv_tend(C.c_int(nchunks), ip(ncols), C.c_double(synth.ZTODT), C.c_int(abi.DYCORE_FV), C.c_double(1.0e-12),
                         *[dp(a) for a in ins], *[dp(outs[k]) for k in names_out], ip(jt), ip(maxg), ip(ideep), ip(lengath),
                         C.byref(nbad))
    return outs, jt, maxg, ideep, lengath, nbad.value


_HALT = r"""
import ctypes as C, os, sys
import numpy as np
root = sys.argv[1]
for p in (root, os.path.join(root, "tests")):
    sys.path.insert(0, p)
import cases
import test_fortran_resident_gpu as T
from cam_nor_physics_amd import synth
from cam_nor_physics_amd.engine import ColumnPhysics
pver, ncol = 32, 16 * 6 - 3
eng = ColumnPhysics(cases.default_params(pver), 0)          # first: the process's HIP runtime is the one torch brings
lib = C.CDLL(os.path.join(root, "cam-nor-physics_amd", "_build", "libcolphys_shim_p16_l32.so"))
st = synth.make_state(ncol, pver, 5)
if sys.argv[2] == "nan":
    st["t"] = st["t"].copy(); st["t"][37, :] = np.nan
print("NBAD", T.call_res_zm_conv_tend(lib, st, ncol, pver)[-1])
"""


@pytest.mark.parametrize("case,halts", [("clean", False), ("nan", True)])
def test_fortran_whole_node_status_check_halts_like_the_reference(case, halts):
    """colphys_check_zm_status (colphys_resident.F90) on the status words of the whole rank: a sounding that is not a number
    ends the run with the reference's ienthalpy message (zm_conv.F90:5565; endrun here is the stand-in that stops the
    process, hence the child process); clean soundings return."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _HALT, ROOT, case], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    if halts:
        assert r.returncode != 0 and "NBAD" not in r.stdout, out[-2000:]
        assert "ENDRUN:" in out and "**** ZM_CONV IENTHALPY: Tmix did not converge ****" in out, out[-2000:]
    else:
        assert r.returncode == 0 and "NBAD 0" in r.stdout and "ENDRUN:" not in out, out[-2000:]


@pytest.mark.parametrize("ncol", [16 * 40, 16 * 40 - 6])
def test_fortran_whole_node_zm_conv_tend_matches_python_engine(ncol):
    from test_parity_gpu import engine
    pver = 32
    lib = C.CDLL(os.path.join(ROOT, "cam-nor-physics_amd", "_build", f"libcolphys_shim_p{PCOLS}_l{pver}.so"))
    params = cases.default_params(pver)
    eng = engine(params)                                    # colphys_init: the Fortran caller shares the engine instance
    st = synth.make_state(ncol, pver, 515)
    d = eng.dev
    S = {k: d(st[k]) for k in ("t", "q", "u", "v", "pint", "pmid", "pdel", "rpdel", "lnpint", "lnpmid", "pdeldry", "phis",
                               "pblh", "tpert", "landfrac", "cld", "rair", "zvir", "zi", "zm")}
    S["ps"] = d(st["pint"][:, -1].copy())
    want = {k: v.cpu().numpy() for k, v in eng.zm_conv_tend(S, synth.ZTODT, pcols=PCOLS).items()}

    nchunks = (ncol + PCOLS - 1) // PCOLS
    outs, jt, maxg, ideep, lengath, nbad = call_res_zm_conv_tend(lib, st, ncol, pver)
    assert nbad == 0
    for k in ("ptend_s", "ptend_q", "ptend_u", "ptend_v", "mconzm", "evapcdp"):
        assert np.array_equal(from_cam(outs[k], ncol), want[k]), k
    for k in ("prec", "snow", "rliq", "jctop", "jcbot"):
        assert np.array_equal(from_cam(outs[k], ncol)[:, 0], want[k]), k
    assert np.array_equal(lengath, want["lengath"]) and lengath.sum() > 0
    for c in range(nchunks):
        n = lengath[c]
        assert np.array_equal(ideep[c, :n], want["ideep"][c * PCOLS:c * PCOLS + n])
        assert np.array_equal(jt[c, :n], want["jt"][c * PCOLS:c * PCOLS + n])
        assert np.array_equal(maxg[c, :n], want["maxg"][c * PCOLS:c * PCOLS + n])
        assert np.array_equal(outs["mu"][c, :, :n].T, want["mu"][c * PCOLS:c * PCOLS + n])
    # padding columns of the host arrays were never written (CAM's +inf / here NaN stays)
    if ncol % PCOLS:
        assert np.isnan(outs["ptend_s"][-1, :, ncol % PCOLS:]).all()


def to_cam_multi(a, nchunks):
    """[nf][ncol][nlev] -> C-ordered (nchunks, nf, nlev, pcols) = Fortran (pcols, nlev, nf, nchunks)"""
    nf, ncol, nlev = a.shape
    out = np.full((nchunks, nf, nlev, PCOLS), np.inf)
    for c in range(nchunks):
        n = min(PCOLS, ncol - c * PCOLS)
        out[c, :, :, :n] = np.transpose(a[:, c * PCOLS:c * PCOLS + n, :], (0, 2, 1))
    return out


@pytest.mark.parametrize("ncol", [16 * 30, 16 * 30 - 5])
def test_fortran_whole_node_optics_match_python_engine(ncol):
    """modal_aero_sw (+ its history diagnostics through the hand-over buffer) and modal_aero_lw of all chunks of a rank from a
    Fortran caller on resident data (tests/fortran/resident_driver.F90: res_modal_aero), bit for bit what the Python front
    end gets from the same C ABI on the dense layout"""
    from test_parity_gpu import engine
    pver = 32
    lib = C.CDLL(os.path.join(ROOT, "cam-nor-physics_amd", "_build", f"libcolphys_shim_p{PCOLS}_l{pver}.so"))
    eng = engine(cases.default_params(pver))
    props = synth.make_aero_props(91)
    eng.aero_tables_set(props, synth.PHYSCONST)
    st = synth.make_state(ncol, pver, 91)
    aer = synth.make_aero_state(ncol, pver, props, 91)
    nm, ns, nsw, nlw = props["nmodes"], props["nspec_max"], synth.NSWBANDS, synth.NLWBANDS
    d = eng.dev
    a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
    vis = eng.empty(3, nm, ncol, pver)
    sw = eng.modal_aero_sw(*a, vis=vis)
    dg = eng.modal_aero_sw_diag(a[0], d(st["pmid"]), d(st["t"]), a[1], a[2], a[3], d(aer["troplevchem"].astype(np.int32)), sw["tauxar"], vis=vis)
    lw = eng.modal_aero_lw(*a)
    want = {k: sw[k].cpu().numpy() for k in ("tauxar", "wa", "ga", "fa")}
    want_lw = lw["tauxar"].cpu().numpy()
    want_lev, want_col = dg["lev"].cpu().numpy(), dg["col"].cpu().numpy()

    nchunks = (ncol + PCOLS - 1) // PCOLS
    ncols = np.array([min(PCOLS, ncol - c * PCOLS) for c in range(nchunks)], dtype=np.int32)
    ins = [to_cam(st[k], nchunks) for k in ("pdeldry", "pmid", "t")]
    mmr = to_cam_multi(aer["specmmr"].reshape(nm * ns, ncol, pver), nchunks)
    dgn = to_cam_multi(aer["dgnumwet"], nchunks)
    qw = to_cam_multi(aer["qaerwat"], nchunks)
    tl = np.ones((nchunks, PCOLS), dtype=np.int32)
    for c in range(nchunks):
        tl[c, :ncols[c]] = aer["troplevchem"][c * PCOLS:c * PCOLS + ncols[c]]
    outs = [np.full((nchunks, (pver + 1) * nsw, PCOLS), np.nan) for _ in range(4)]
    lwtau = np.full((nchunks, pver * nlw, PCOLS), np.nan)
    ncf = abi.SWD_NCOL + abi.SWD_NMODEF * nm
    dlev = np.full((abi.SWD_NLEV, nchunks, pver, PCOLS), np.nan)
    dcol = np.full((ncf, nchunks, 1, PCOLS), np.nan)
    nbad = C.c_int(-7)
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))
    lib.res_modal_aero(C.c_int(nchunks), ip(ncols), C.c_int(nm), C.c_int(ns), C.c_int(nsw), C.c_int(nlw), *[dp(x) for x in ins],
                       dp(mmr), dp(dgn), dp(qw), ip(tl), *[dp(x) for x in outs], dp(lwtau), dp(dlev), dp(dcol), C.byref(nbad))
    assert nbad.value == 0
    for k, o in zip(("tauxar", "wa", "ga", "fa"), outs):
        assert np.array_equal(from_cam(o, ncol).reshape(ncol, nsw, pver + 1), want[k]), k
    assert np.array_equal(from_cam(lwtau, ncol).reshape(ncol, nlw, pver), want_lw)
    for f in range(abi.SWD_NLEV):
        assert np.array_equal(from_cam(dlev[f], ncol), want_lev[f]), f
    for f in range(ncf):
        assert np.array_equal(from_cam(dcol[f], ncol)[:, 0], want_col[f]), f
    if ncol % PCOLS:
        assert np.isnan(outs[0][-1, :, ncol % PCOLS:]).all()
