"""The whole-rank HOST FACE (include/colphys.h colphys_hostface_*): chunk records in the reference's (pcols, nlev) layout in
pinned memory -> upload -> device re-layout -> zm_conv_tend (+ optics) -> device re-layout -> download, pipelined over slices
on four streams (upload, two for kernels, download), against the RESIDENT path on the same columns: bit for bit, per-column fields and gathered sets alike."""
import numpy as np
import pytest

import cases
from cam_nor_physics_amd import abi, synth
from cam_nor_physics_amd.hostface import HF_AERO, HF_HIST, HF_SWDIAG, HF_ZM, HostFace, chunk_counts
from test_parity_gpu import engine, to_np

pytestmark = pytest.mark.gpu

ZM_COL = ("prec", "snow", "rliq", "rice", "jctop", "jcbot", "cape", "freqzm", "pcont", "pconb")
ZM_MID = ("ptend_s", "ptend_q", "ptend_u", "ptend_v", "ql", "rprd", "evapcdp", "dlf")
ZM_INT = ("mconzm", "flxprec", "flxsnow")
ZM_GATH = ("mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg", "ideep")
HIST_1 = ("cme", "zdu", "dif", "eurt", "zmdt_s", "zmdq", "evap_s", "evap_s_snwprd", "evap_s_snwevmlt", "ntprprd", "ntsnprd", "seten", "pflx")
HIST_2 = ("pguall", "pgdall", "icwu", "icwd")


def resident(eng, st, aer, props, pcols, sw_diag, hist=False):
    d = eng.dev
    S = {k: d(st[k]) for k in abi.ZM_TEND_IN if k != "ps"}
    S["ps"] = d(np.ascontiguousarray(st["pint"][:, -1]))
    o = to_np(eng.zm_conv_tend(S, synth.ZTODT, pcols=pcols, hist=hist))
    if aer is not None:
        n, pver = st["t"].shape
        a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
        vis = eng.empty(3, eng.tables.nmodes, n, pver) if sw_diag else None
        sw = eng.modal_aero_sw(*a, vis=vis)
        if sw_diag:
            dg = eng.modal_aero_sw_diag(a[0], d(st["pmid"]), d(st["t"]), a[1], a[2], a[3], d(aer["troplevchem"].astype(np.int32)), sw["tauxar"], vis=vis)
            o.update(swd_lev=dg["lev"].cpu().numpy(), swd_col=dg["col"].cpu().numpy())
        lw = eng.modal_aero_lw(*a)
        o.update({"sw_" + k: v.cpu().numpy() for k, v in sw.items()})
        o.update(lw_tauxar=lw["tauxar"].cpu().numpy(), lw_status=lw["status"].cpu().numpy())
    return o


def active_rows(ncols, pcols):
    """row index in the padded [nchunks*pcols] layout of every active column, in order"""
    return np.concatenate([c * pcols + np.arange(n) for c, n in enumerate(ncols)])


def compare(hf, r, ncols, pcols, pver, with_aero, sw_diag, hist, same_chunking, with_zm=True):
    ncol = int(np.sum(ncols))
    if with_zm:
        compare_zm(hf, r, ncols, pcols, pver, hist, same_chunking)
    if with_aero:
        compare_optics(hf, r, ncol, pver, sw_diag)


def compare_zm(hf, r, ncols, pcols, pver, hist, same_chunking):
    ncol = int(np.sum(ncols))
    for k in ZM_COL + ZM_MID + ZM_INT:
        assert np.array_equal(hf.get_columns(k), r[k]), k
    assert np.array_equal(hf.get_columns("status"), r["status"])
    if hist:
        for k in HIST_1:
            assert np.array_equal(hf.get_columns(k), r[k]), k
        for k in HIST_2:
            assert np.array_equal(hf.get_columns(k + "_u"), r[k][0]) and np.array_equal(hf.get_columns(k + "_v"), r[k][1]), k
    lg = hf.view("lengath").copy()
    if same_chunking:                       # only the last chunk is ragged: the resident run has the very same chunks
        assert np.array_equal(lg, r["lengath"])
        for k in ZM_GATH:
            assert np.array_equal(hf.get_columns(k, gathered=True)[:ncol], r[k]), k
    # the gathered sets are a faithful ascending compaction of the ACTIVE deep columns, shadows dropped
    rows = active_rows(ncols, pcols)
    deep_cols = np.flatnonzero(r["freqzm"] > 0)
    assert lg.sum() == len(deep_cols)
    ideep = hf.get_columns("ideep", gathered=True)
    jt, maxg, mu = hf.get_columns("jt", gathered=True), hf.get_columns("maxg", gathered=True), hf.get_columns("mu", gathered=True)
    src = []
    for c, n in enumerate(ncols):
        ids = ideep[c * pcols:c * pcols + lg[c]]
        assert np.all(np.diff(ids) > 0) and (ids >= 1).all() and (ids <= n).all()
        assert not ideep[c * pcols + lg[c]:(c + 1) * pcols].any()
        src.append(c * pcols + ids - 1)
    src = np.concatenate(src) if src else np.zeros(0, int)
    col_of_row = -np.ones(len(ncols) * pcols, int); col_of_row[rows] = np.arange(ncol)
    assert np.array_equal(np.sort(col_of_row[src]), deep_cols)
    g_rows = np.concatenate([c * pcols + np.arange(lg[c]) for c in range(len(ncols))]).astype(int)
    assert np.array_equal(jt[g_rows], r["jctop"][col_of_row[src]].astype(np.int32))
    assert np.array_equal(maxg[g_rows], r["jcbot"][col_of_row[src]].astype(np.int32))
    assert np.array_equal(mu[g_rows] * 100.0 / synth.PHYSCONST["gravit"], r["mu_out"][col_of_row[src]])


def compare_optics(hf, r, ncol, pver, sw_diag):
    nsw, nlw = synth.NSWBANDS, synth.NLWBANDS
    for k in ("tauxar", "wa", "ga", "fa"):
        got = hf.get_columns("sw_" + k).reshape(ncol, nsw, pver + 1)              # (pcols, 0:pver, band): level fastest, then band
        assert np.array_equal(got, r["sw_" + k]), k
    assert np.array_equal(hf.get_columns("lw_tauxar").reshape(ncol, nlw, pver), r["lw_tauxar"])
    assert np.array_equal(hf.get_columns("sw_status"), r["sw_status"]) and np.array_equal(hf.get_columns("lw_status"), r["lw_status"])
    if sw_diag:
        lev = hf.get_columns("swd_lev").reshape(ncol, abi.SWD_NLEV, pver).transpose(1, 0, 2)
        col = hf.get_columns("swd_col").T
        assert np.array_equal(lev, r["swd_lev"]) and np.array_equal(col, r["swd_col"])


@pytest.mark.parametrize("pver,ncols,what,nslices,pcols", [
    (32, [16, 16, 9, 16, 1, 16, 16, 3], HF_ZM | HF_AERO | HF_SWDIAG | HF_HIST, 3, 16),   # ragged chunks anywhere, three slices
    (32, [16] * 20 + [7], HF_ZM, 1, 16),                                                   # ZM only, one slice
    (70, [16] * 9 + [12], HF_ZM | HF_AERO, 2, 16),                                         # L70: wide optics records (14 x 71 levels) in pieces
    (26, [5], HF_ZM | HF_AERO | HF_SWDIAG, 0, 16),                                         # one small chunk
    (32, [48, 33, 48, 48, 17], HF_ZM | HF_AERO | HF_SWDIAG | HF_HIST, 2, 48),             # chunks wider than a re-layout block (two column blocks)
    (32, [8] * 30 + [3], HF_ZM | HF_AERO, 4, 8),                                           # narrow chunks
    (32, [16] * 11 + [5], HF_AERO | HF_SWDIAG, 3, 16),                                     # the optics alone (the radiation driver's call)
    (70, [16, 2], HF_AERO, 0, 16),
])
def test_hostface_matches_resident_path(pver, ncols, what, nslices, pcols):
    ncols = np.array(ncols, np.int32)
    ncol = int(ncols.sum())
    eng = engine(cases.default_params(pver))
    with_zm, with_aero, sw_diag, hist = bool(what & HF_ZM), bool(what & HF_AERO), bool(what & HF_SWDIAG), bool(what & HF_HIST)
    props = synth.make_aero_props(3)
    st = synth.make_state(ncol, pver, 1212, moist_boost=1.15)
    aer = synth.make_aero_state(ncol, pver, props, 3) if with_aero else None
    if with_aero:
        eng.aero_tables_set(props, synth.PHYSCONST)
    r = resident(eng, st, aer, props, pcols, sw_diag, hist)
    hf = HostFace(eng, ncols, pcols, pver, what, nslices)
    try:
        hf.fill_from_synth(st, aer, props)
        t = hf.run(synth.ZTODT)
        assert t["wall_ms"] > 0 and t["h2d_bytes"] == len(ncols) * hf.in_rec and t["d2h_bytes"] == len(ncols) * hf.out_rec
        same = bool((ncols[:-1] == pcols).all())
        compare(hf, r, ncols, pcols, pver, with_aero, sw_diag, hist, same, with_zm)
        if not with_zm:
            assert "lengath" not in hf.fields and "ptend_s" not in hf.fields and "q" not in hf.fields
        # a second run on the same records gives the same bits (no state is kept between runs)
        first = {k: hf.view(k).copy() for k in (("ptend_s", "mu", "ideep", "lengath") if with_zm else ("sw_tauxar", "lw_tauxar"))}
        hf.run(synth.ZTODT)
        for k, v in first.items():
            assert np.array_equal(hf.view(k), v), k
    finally:
        hf.close()
    if pver >= 26 and ncol > 100:
        assert r["lengath"].sum() > 0


def test_hostface_argument_errors():
    eng = engine(cases.default_params(32))
    eng.aero_tables_set(synth.make_aero_props(3), synth.PHYSCONST)
    ncols = np.array([16, 4], np.int32)
    for what in (0, HF_SWDIAG, HF_HIST | HF_AERO, HF_SWDIAG | HF_ZM):
        with pytest.raises(RuntimeError):
            HostFace(eng, ncols, 16, 32, what)


def test_hostface_ne30_rank_matches_resident_path_and_reports_its_times():
    """48 602 columns = 3 037 chunks of 16 and one of 10, eight slices on four streams, ZM + optics + SW diagnostics: every
    output of every active column and the gathered sets of every chunk equal the resident path's; the host's share of a run is
    enqueueing"""
    pver, pcols, ncol = 32, 16, 48602
    eng = engine(cases.default_params(pver))
    props = synth.make_aero_props(synth.DEFAULT_SEED)
    st = synth.make_state(ncol, pver, synth.DEFAULT_SEED, moist_boost=1.1)
    aer = synth.make_aero_state(ncol, pver, props, synth.DEFAULT_SEED)
    eng.aero_tables_set(props, synth.PHYSCONST)
    r = resident(eng, st, aer, props, pcols, True)
    ncols = chunk_counts(ncol, pcols)
    hf = HostFace(eng, ncols, pcols, pver, HF_ZM | HF_AERO | HF_SWDIAG)
    try:
        hf.fill_from_synth(st, aer, props)
        hf.run(synth.ZTODT)
        t = hf.run(synth.ZTODT)
        compare(hf, r, ncols, pcols, pver, True, True, False, True)
        assert t["nslices"] == 8 and t["nstreams"] == 4
        assert t["host_cpu_ms"] < 5.0, t                 # enqueueing ~60 launches and 16 copies; the rest of the call sleeps
        assert t["wall_ms"] < 200.0, t
    finally:
        hf.close()
    eng.torch.cuda.empty_cache()
