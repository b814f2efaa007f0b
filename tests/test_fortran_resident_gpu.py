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
    assert nbad.value == 0
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
