"""GPU parity tests: the HIP path, called through the C ABI (libcolphys_hip.so), against
  (a) the golden fixtures generated from the reference Fortran itself (tests/golden/make_golden.py),
  (b) the C restatement (oracle/colphys_oracle.c) on fresh seeded inputs,
  (c) size-independent properties at BASELINE.json's full size (48 602 columns): column independence,
      chunk-width independence, gathered-layout consistency.
Tolerance 1e-10 relative (tests/util.py); integer outputs identical.
"""
import os

import numpy as np
import pytest

import cases
from cam_nor_physics_amd import abi, synth
from util import assert_close, gathered_mask

pytestmark = pytest.mark.gpu

_engines = {}


def engine(params):
    """one engine per process; colphys_init may be called again to change parameters"""
    from cam_nor_physics_amd.engine import ColumnPhysics
    eng = _engines.get("e")
    if eng is None:
        eng = ColumnPhysics(params, 0)
        _engines["e"] = eng
    else:
        eng._check(eng.lib.colphys_init(0, params))
        eng.params = params
    return eng


def to_np(o):
    return {k: v.cpu().numpy() for k, v in o.items()}


def run_zm(eng, args, pcols=cases.PCOLS):
    t, qh, pblh, zm, geos, zi, pap, paph, dpp, delt, tpert, landfrac = args
    d = eng.dev
    o = eng.zm_convr(d(t), d(qh), d(pblh), d(zm), d(geos), d(zi), d(pap), d(paph), d(dpp), delt, d(tpert), d(landfrac), pcols=pcols)
    eng.torch.cuda.synchronize()
    return to_np(o)


def check_zm(o, ref, ncol, pcols=cases.PCOLS):
    assert np.array_equal(o["lengath"], ref["lengath"]), (o["lengath"], ref["lengath"])
    m = gathered_mask(ref["lengath"], ncol, pcols)
    worst = 0.0
    for k in abi.ZM_CONVR_OUT_MID + abi.ZM_CONVR_OUT_INT + abi.ZM_CONVR_OUT_COL + ("jt", "maxg", "ideep"):
        a, b = o[k], ref[k]
        if k in abi.ZM_CONVR_GATHERED or k == "ideep":
            assert not np.any(a[~m]), f"{k}: slots beyond lengath must be zero"
            a, b = a[m], b[m]
        worst = max(worst, assert_close("zm_convr." + k, a, b))
    assert not o["status"].any()
    return worst


# ---------------------------------------------------------------------------------------------------
# (a) golden fixtures = outputs of the reference Fortran
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("trig_mode", ["auto", "lane-pairs", "one-lane", "fused", "plume-pairs"])
@pytest.mark.parametrize("name", cases.ZM_GOLDEN)
def test_zm_convr_vs_reference_golden(name, trig_mode, monkeypatch):
    """both mappings of the first ascent: wave pairs (what these column counts select) and the single-wave kernel
    that takes over when the column groups fill the SIMDs (forced here through COLPHYS_TRIG_MODE); and the plume kernels with
    two columns per wave, the plume kernels' other mapping at up to 32 levels (forced through COLPHYS_PLUME_CPW; L70 keeps
    one column per wave whatever the setting)"""
    if trig_mode == "one-lane":                       # the wave-pair trigger with one lane per column (what > 16 384 columns select;
        monkeypatch.setenv("COLPHYS_LPC_TRIG", "1")   # these column counts take four lanes per column by themselves: "auto")
        monkeypatch.setenv("COLPHYS_LPC_ASC", "2")
    if trig_mode == "lane-pairs":                     # two lanes per column in both ascents (8 192 to 16 384 columns / beyond 24 000)
        monkeypatch.setenv("COLPHYS_LPC_TRIG", "2")
        monkeypatch.setenv("COLPHYS_LPC_ASC", "2")
    if trig_mode == "fused":
        monkeypatch.setenv("COLPHYS_TRIG_MODE", "2")
    if trig_mode == "plume-pairs":
        monkeypatch.setenv("COLPHYS_PLUME_CPW", "2")
    g = cases.load_golden(name)
    eng = engine(cases.golden_params(g))
    o = run_zm(eng, cases.zm_args_from_golden(g))
    ref = {k[3:]: v for k, v in g.items() if k.startswith("zm_")}
    check_zm(o, ref, g["in_t"].shape[0])
    assert ref["lengath"].sum() > 0


@pytest.mark.parametrize("name", cases.ZM_GOLDEN)
def test_geopotential_vs_reference_golden(name):
    g = cases.load_golden(name)
    eng = engine(cases.golden_params(g))
    d = eng.dev
    q2 = np.stack([g["in_q"], 0.1 * g["in_q"]])
    for dy, tag in ((abi.DYCORE_EUL, "eul"), (abi.DYCORE_FV, "fv"), (abi.DYCORE_SE, "se")):
        zi, zm = eng.geopotential_t(d(g["in_lnpint"]), d(g["in_lnpmid"]), d(g["in_pint"]), d(g["in_pmid"]), d(g["in_pdel"]),
                                    d(g["in_rpdel"]), d(g["in_t"]), d(q2), d(g["in_rair"]), synth.PHYSCONST["gravit"],
                                    d(g["in_zvir"]), dycore=dy, nspecies=2 if dy == abi.DYCORE_SE else 1)
        assert_close(f"geopotential_t[{tag}].zi", zi.cpu().numpy(), g[f"geo_{tag}_zi"])
        assert_close(f"geopotential_t[{tag}].zm", zm.cpu().numpy(), g[f"geo_{tag}_zm"])


@pytest.mark.parametrize("pver", [32, 70])
def test_geopotential_dse_vs_reference_golden(pver):
    g = cases.load_golden(f"geo_dse_l{pver}")
    eng = engine(cases.default_params(pver))
    d = eng.dev
    for dy, tag in ((abi.DYCORE_EUL, "eul"), (abi.DYCORE_FV, "fv"), (abi.DYCORE_SE, "se")):
        args, nsp, t_true = cases.geo_dse_args_from_golden(g, tag, int(g["ncol"]), pver, int(g["seed"]), dy)
        a = [d(x) if isinstance(x, np.ndarray) else x for x in args]
        t, zi, zm = eng.geopotential_dse(*a, dycore=dy, nspecies=nsp)
        # the same IEEE operations in the same order: bit-exact, like geopotential_t
        assert np.array_equal(t.cpu().numpy(), g[f"{tag}_t"]), tag
        assert np.array_equal(zi.cpu().numpy(), g[f"{tag}_zi"]) and np.array_equal(zm.cpu().numpy(), g[f"{tag}_zm"]), tag


def test_geopotential_dse_full_size_round_trip():
    """48 602 columns (ragged last tile): T -> dse with the engine's own geopotential_t heights -> geopotential_dse gives the
    temperature and the heights back (size-independent property)."""
    pver, ncol = 32, 48602
    eng = engine(cases.default_params(pver))
    d = eng.dev
    st = synth.make_state(ncol, pver, 99)
    c = synth.PHYSCONST
    S = {k: d(st[k]) for k in ("lnpint", "lnpmid", "pint", "pmid", "pdel", "rpdel", "t", "q", "rair", "zvir", "phis")}
    zi, zm = eng.geopotential_t(S["lnpint"], S["lnpmid"], S["pint"], S["pmid"], S["pdel"], S["rpdel"], S["t"], S["q"], S["rair"],
                                c["gravit"], S["zvir"])
    cpair = eng.torch.full_like(S["t"], c["cpair"])
    dse = cpair * S["t"] + c["gravit"] * zm + S["phis"][:, None]
    t2, zi2, zm2 = eng.geopotential_dse(S["lnpint"], S["lnpmid"], S["pint"], S["pmid"], S["pdel"], S["rpdel"], dse, S["q"], S["phis"],
                                        S["rair"], c["gravit"], cpair, S["zvir"])
    assert float(((t2 - S["t"]).abs() / S["t"]).max()) < 1e-12
    assert float(((zi2 - zi).abs().max(dim=1).values / zi.abs().max(dim=1).values).max()) < 1e-12
    assert float(((zm2 - zm).abs().max(dim=1).values / zm.abs().max(dim=1).values).max()) < 1e-12


@pytest.mark.parametrize("ncol,pver,ncnst,dycore,nspecies", [(1003, 32, 5, abi.DYCORE_FV, 1), (48602, 32, 3, abi.DYCORE_FV, 1),
                                                             (333, 70, 7, abi.DYCORE_SE, 2), (64, 26, 300, abi.DYCORE_EUL, 1)])
def test_derived_fields_vs_c_oracle(ncol, pver, ncnst, dycore, nspecies):
    """d_p_coupling's derived fields on resident data: pressures, bottom-layer fix, heights and dry static energy are the
    oracle's bit for bit; log(pmid) is the device logarithm (< 1 ulp class): 1e-10 relative as everywhere else."""
    from c_oracle import COracle
    a = cases.derived_fields_case(ncol, pver, ncnst, 11 + ncol)
    params = cases.default_params(pver)
    qw = a["q"].copy()
    want = COracle(params).derived_fields(a["pint"], a["lnpint"], a["t"], qw, a["qmin"], a["phis"], a["cpair"], a["rair"], a["zvir"],
                                          a["gravit"], dycore=dycore, nspecies=nspecies)
    eng = engine(params)
    d = eng.dev
    qg = d(a["q"])
    got = eng.derived_fields(d(a["pint"]), d(a["lnpint"]), d(a["t"]), qg, a["qmin"], d(a["phis"]), d(a["cpair"]), d(a["rair"]), d(a["zvir"]),
                             a["gravit"], dycore=dycore, nspecies=nspecies)
    assert np.array_equal(qg.cpu().numpy(), qw)
    for k in ("pdel", "rpdel", "pmid", "zi", "zm", "s"):
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    assert_close("lnpmid", got["lnpmid"].cpu().numpy(), want["lnpmid"])


@pytest.mark.parametrize("clubb", [True, False])
@pytest.mark.parametrize("ncol,pver", [(1003, 32), (48602, 32), (257, 70)])
def test_convect_diagnostics_vs_c_oracle(ncol, pver, clubb):
    from c_oracle import COracle
    f = cases.convect_diag_case(ncol, pver, 8 + ncol)
    params = cases.default_params(pver)
    want = COracle(params).convect_diagnostics({k: v.copy() for k, v in f.items()}, clubb)
    eng = engine(params)
    got = eng.convect_diagnostics({k: eng.dev(v) for k, v in f.items()}, clubb)
    for k in want:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    if clubb:                                        # the CLUBB branch accepts a caller without any shallow-scheme field
        sub = {k: eng.dev(v) for k, v in f.items() if k in ("cmfmc", "qc", "rliq", "cnt", "cnb", "rprddp", "pmid", "rprdtot", "pcnt", "pcnb")}
        got2 = eng.convect_diagnostics(sub, True)
        for k in sub:
            assert np.array_equal(got2[k].cpu().numpy(), want[k]), k


@pytest.mark.parametrize("name", cases.ZM_GOLDEN)
def test_evap_and_momtran_vs_reference_golden(name):
    g = cases.load_golden(name)
    eng = engine(cases.golden_params(g))
    d = eng.dev
    prec = d(g["zm_prec"].copy())
    e = to_np(eng.zm_conv_evap(d(g["in_evap_t"]), d(g["in_pmid"]), d(g["in_pdel"]), d(g["in_evap_q"]), d(g["in_landfrac"]),
                               d(g["zm_rprd"]), d(g["in_cld"]), synth.ZTODT, prec))
    for k, v in e.items():
        assert_close("zm_conv_evap." + k, v, g["evap_" + k])
    winds = d(np.stack([g["in_u"], g["in_v"]]))
    m = to_np(eng.momtran(winds, d(g["zm_mu"]), d(g["zm_md"]), d(g["zm_du"]), d(g["zm_eu"]), d(g["zm_ed"]), d(g["zm_dp"]),
                          d(g["zm_dsubcld"]), d(g["zm_jt"]), d(g["zm_maxg"]), d(g["zm_ideep"]), d(g["zm_lengath"]), synth.ZTODT))
    for k, v in m.items():
        assert_close("momtran." + k, v, g["mom_" + k])
    cq, cf, cdp = cases.convtran_inputs(g["in_pdel"], g["in_q"], g["zm_ideep"], g["zm_lengath"], int(g["ct_seed"]))
    ct = eng.convtran(d(cq), d(cf), d(g["zm_mu"]), d(g["zm_md"]), d(g["zm_du"]), d(g["zm_eu"]), d(g["zm_ed"]), d(g["zm_dp"]), d(cdp),
                      d(g["zm_jt"]), d(g["zm_maxg"]), d(g["zm_ideep"]), d(g["zm_lengath"]), cases.DOCONVTRAN, cases.ISDRY, synth.ZTODT)
    assert_close("convtran.dqdt", ct.cpu().numpy(), g["ct_dqdt"])
    # zm_conv_tend_2 on resident data (zm_conv_intr.F90:955-1028): the same transport with lengath = count(ideep > 0) and
    # dpdry gathered from state%pdeldry on the fly -- the reference's convtran output for these inputs is the golden
    pdeldry = g["in_pdel"] * (1.0 - g["in_q"])
    t2 = eng.zm_conv_tend_2(d(cq), d(cf), d(pdeldry), d(g["zm_mu"]), d(g["zm_md"]), d(g["zm_du"]), d(g["zm_eu"]), d(g["zm_ed"]),
                            d(g["zm_dp"]), d(g["zm_jt"]), d(g["zm_maxg"]), d(g["zm_ideep"]), cases.DOCONVTRAN, cases.ISDRY, synth.ZTODT)
    assert_close("zm_conv_tend_2.ptend_q", t2.cpu().numpy(), g["ct_dqdt"])


@pytest.mark.parametrize("ncnst,pver,ncol", [(3, 32, 640), (41, 32, 2000), (70, 32, 1003), (131, 26, 400), (9, 70, 333), (6, 128, 300),
                                             (4, 12, 900)])
def test_convtran_many_tracers_vs_c_oracle(ncnst, pver, ncol):
    """One launch for all transported tracers (lane = tracer, groups of 64, launches of 120): every grouping boundary,
    dry and moist constituents mixed, ragged last chunk, against the C restatement; untouched tracers stay untouched."""
    from c_oracle import COracle
    params = cases.default_params(pver)
    eng, co = engine(params), COracle(params)
    d = eng.dev
    st = synth.make_state(ncol, pver, 4242 + ncnst)
    zi, zm = co.geopotential_t(*cases.geo_args(st), st["q"], st["rair"], synth.PHYSCONST["gravit"], st["zvir"])
    z = co.zm_convr(*cases.zm_args_from_state(st, zi, zm))
    assert int(z["lengath"].sum()) > 0
    rng = np.random.default_rng(ncnst)
    lev = np.linspace(0.0, 1.0, pver)[None, None, :]
    q = rng.uniform(0.2, 1.0, (ncnst, ncol, 1)) * 1.0e-6 * (0.05 + lev ** rng.uniform(0.5, 3.0, (ncnst, ncol, 1)))
    q *= 1.0 + 0.2 * rng.standard_normal((ncnst, ncol, pver))
    q[1 % ncnst, :, : pver // 3] = 0.0
    q[2 % ncnst, ::5, pver // 2] *= -1.0
    fracis = rng.uniform(0.0, 1.0, (ncnst, ncol, pver))
    lq = (rng.uniform(size=ncnst) < 0.8).astype(np.int32); lq[0] = 0; lq[-1] = 1
    isdry = (rng.uniform(size=ncnst) < 0.4).astype(np.int32)
    pdeldry = st["pdel"] * (1.0 - st["q"])
    dpdry = np.ones((ncol, pver))
    for c in range(len(z["lengath"])):
        for j in range(int(z["lengath"][c])):
            row = c * cases.PCOLS + j
            dpdry[row] = pdeldry[c * cases.PCOLS + int(z["ideep"][row]) - 1] / 100.0
    ref = co.convtran(q, fracis, z["mu"], z["md"], z["du"], z["eu"], z["ed"], z["dp"], dpdry, z["jt"], z["maxg"], z["ideep"],
                      z["lengath"], lq, isdry, synth.ZTODT)
    sentinel = 7.0
    out = eng.torch.full((ncnst, ncol, pver), sentinel, dtype=eng.torch.float64, device=eng.device)
    got = eng.zm_conv_tend_2(d(q), d(fracis), d(pdeldry), d(z["mu"]), d(z["md"]), d(z["du"]), d(z["eu"]), d(z["ed"]), d(z["dp"]),
                             d(z["jt"]), d(z["maxg"]), d(z["ideep"]), lq, isdry, synth.ZTODT, ptend_q=out).cpu().numpy()
    for m in range(ncnst):
        if lq[m] and m > 0:
            assert_close(f"ptend_q[{m}]", got[m], ref[m])
        else:
            assert np.all(got[m] == sentinel), m                    # not selected: left untouched (zm_conv.F90:2299 is inside the if)
    got2 = eng.convtran(d(q), d(fracis), d(z["mu"]), d(z["md"]), d(z["du"]), d(z["eu"]), d(z["ed"]), d(z["dp"]), d(dpdry), d(z["jt"]),
                        d(z["maxg"]), d(z["ideep"]), d(z["lengath"]), lq, isdry, synth.ZTODT).cpu().numpy()
    for m in range(1, ncnst):
        if lq[m]:
            assert np.array_equal(got2[m], got[m]), m               # gathered dpdry argument == pdeldry gathered on the fly


@pytest.mark.parametrize("family", ["bands-outer", "grouped"])
@pytest.mark.parametrize("name", cases.AERO_GOLDEN)
def test_modal_aero_vs_reference_golden(name, family):
    """both kernel families, SW and LW (colphys_aero_kernel_select): bands-outer (up to four modes) and the grouped one that
    also serves more modes"""
    g = cases.load_golden(name)
    pver = g["in_t"].shape[1]
    eng = engine(cases.default_params(pver))
    eng.aero_tables_set(cases.aero_props_from_golden(g), synth.PHYSCONST)
    d = eng.dev
    a = (d(g["in_pdeldry"]), d(g["in_specmmr"]), d(g["in_dgnumwet"]), d(g["in_qaerwat"]))
    eng.aero_kernel_select(family, family)
    try:
        sw = to_np(eng.modal_aero_sw(*a))
        lw = to_np(eng.modal_aero_lw(*a))
    finally:
        eng.aero_kernel_select("auto", "auto")
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, sw[k], g["sw_" + k])
    assert_close("modal_aero_lw.tauxar", lw["tauxar"], g["lw_tauxar"])
    assert not sw["status"].any() and not lw["status"].any()
    assert (sw["tauxar"] >= 0).all() and (lw["tauxar"] >= 0).all()


def _swd_engine(eng, st, aer, hand_over=True):
    """modal_aero_sw followed by its history diagnostics on the engine; numpy lev / col.  hand_over: modal_aero_sw leaves the
    visible-band optics per mode for the diagnostics (colphys_modal_aero_io.vis_optics); else they evaluate the tables again"""
    d = eng.dev
    a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
    n, pver = st["pdeldry"].shape
    vis = eng.empty(3, eng.tables.nmodes, n, pver) if hand_over else None
    sw = eng.modal_aero_sw(*a, vis=vis)
    dg = eng.modal_aero_sw_diag(a[0], d(st["pmid"]), d(st["t"]), a[1], a[2], a[3], d(aer["troplevchem"].astype(np.int32)), sw["tauxar"],
                                vis=vis)
    return to_np(dg)


def _assert_swd(got, want_lev, want_col, nmodes):
    lev_names, col_names = cases.swd_field_names(nmodes)
    for f, nm in enumerate(lev_names):
        assert_close(nm, got["lev"][f], want_lev[f])
    for f, nm in enumerate(col_names):
        assert_close(nm, got["col"][f], want_col[f])


@pytest.mark.parametrize("hand_over", [True, False])
@pytest.mark.parametrize("name", cases.AERO_GOLDEN)
def test_modal_aero_sw_diagnostics_vs_reference_golden(name, hand_over):
    """modal_aero_sw's history diagnostics (modal_aer_opt.F90:896-993, 1046-1154) against what the reference build handed to
    outfld ("dn" fields; SSAVISdn finished the reference's way on the caller side, cases.swd_reference_ssavis).
    Tolerance 1e-10 relative (tests/util.py); AODTOT / UV / near-IR come from tauxar, i.e. summed over modes first."""
    g = cases.load_golden(name)
    pver = g["in_t"].shape[1]
    eng = engine(cases.default_params(pver))
    eng.aero_tables_set(cases.aero_props_from_golden(g), synth.PHYSCONST)
    st = dict(pdeldry=g["in_pdeldry"], pmid=g["in_pmid"], t=g["in_t"])
    aer = {k: g["in_" + k] for k in ("specmmr", "dgnumwet", "qaerwat", "troplevchem")}
    dg = _swd_engine(eng, st, aer, hand_over)
    dg["col"] = cases.swd_reference_ssavis(dg["col"], g["in_nite"])
    _assert_swd(dg, g["swd_lev"], g["swd_col"], eng.tables.nmodes)


# ---------------------------------------------------------------------------------------------------
# (b) C restatement on fresh seeded inputs, sizes the oracle finishes in seconds
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pver,ncol,seed", [(32, 3001, 11), (70, 801, 12), (26, 333, 13),
                                            (128, 700, 14), (12, 900, 15), (4, 300, 16)])     # COLPHYS_MAX_PVER and the smallest grids
def test_full_chain_vs_c_oracle(pver, ncol, seed):
    """(also the limits of the level count: 128 = COLPHYS_MAX_PVER, levels spanning two lane strides of the wave-per-column
    kernels; 12 levels, where a handful of columns still convect; 4, the fewest the ABI accepts, where none does)"""
    from c_oracle import COracle
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    st = synth.make_state(ncol, pver, seed)
    c = synth.PHYSCONST
    d = eng.dev
    ga = cases.geo_args(st)
    zi_r, zm_r = co.geopotential_t(*ga, st["q"], st["rair"], c["gravit"], st["zvir"], dycore=abi.DYCORE_FV)
    zi, zm = eng.geopotential_t(*[d(x) for x in ga], d(st["q"]), d(st["rair"]), c["gravit"], d(st["zvir"]))
    assert_close("geopotential_t.zi", zi.cpu().numpy(), zi_r)
    assert_close("geopotential_t.zm", zm.cpu().numpy(), zm_r)
    args = cases.zm_args_from_state(st, zi_r, zm_r)
    ref = co.zm_convr(*args)
    o = run_zm(eng, args)
    check_zm(o, ref, ncol)
    frac = ref["lengath"].sum() / ncol
    assert (0.02 < frac < 0.6) if pver >= 26 else (ref["lengath"].sum() > 0) == (pver >= 12), frac
    # evap on the updated state, then momtran on the gathered fluxes
    dt = synth.ZTODT
    t1 = st["t"] + ref["heat"] / c["cpair"] * dt
    q1 = st["q"] + ref["qtnd"] * dt
    ea = (t1, st["pmid"], st["pdel"], q1, st["landfrac"], ref["rprd"], st["cld"], dt, ref["prec"])
    e_ref = co.zm_conv_evap(*ea)
    e = to_np(eng.zm_conv_evap(*[d(x) for x in ea[:7]], dt, d(ref["prec"].copy())))
    for k, v in e.items():
        assert_close("zm_conv_evap." + k, v, e_ref[k])
    m_ref = co.momtran(st["u"], st["v"], ref["mu"], ref["md"], ref["du"], ref["eu"], ref["ed"], ref["dp"], ref["dsubcld"],
                       ref["jt"], ref["maxg"], ref["ideep"], ref["lengath"], dt)
    m = to_np(eng.momtran(d(np.stack([st["u"], st["v"]])), *[d(ref[k]) for k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld")],
                          *[d(ref[k]) for k in ("jt", "maxg", "ideep", "lengath")], dt))
    for k, v in m.items():
        assert_close("momtran." + k, v, m_ref[k])


def test_namelist_and_sounding_regimes_vs_c_oracle():
    """scripts/stress_parity.py, ten seeded regimes: PBL-mixed parcels, no_deep_pbl, 1..5 CIN regions, tiedke_add, capelmt, tau,
    dmpdz, momentum coefficients, autoconversion and evaporation coefficients, cold / hot / dry / very moist soundings, odd
    level counts -- zm_convr, zm_conv_evap and momtran against the C restatement at 1e-10, gathered sets identical"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_REGIMES="10", STRESS_SEED="7")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_parity.py")], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("worst over")]
    assert last and float(last[-1].split()[-1]) < 1e-10, r.stdout[-2000:]
    assert sum(ln.startswith("regime") for ln in r.stdout.splitlines()) == 10, r.stdout[-2000:]


def test_evap_snow_paths_vs_c_oracle():
    from c_oracle import COracle
    params = cases.default_params(32)
    eng = engine(params)
    co = COracle(params)
    ea = cases.evap_cold_case(1500, 32, 21)
    e_ref = co.zm_conv_evap(*ea)
    assert e_ref["snow"].max() > 0 and (e_ref["tend_s_snwevmlt"] != 0).any()
    d = eng.dev
    e = to_np(eng.zm_conv_evap(*[d(x) for x in ea[:7]], ea[7], d(ea[8].copy())))
    for k, v in e.items():
        assert_close("zm_conv_evap." + k, v, e_ref[k])


def test_evap_snow_paths_vs_reference_golden():
    g = cases.load_golden("evap_cold")
    eng = engine(cases.default_params(int(g["pver"])))
    ea = cases.evap_cold_case(int(g["ncol"]), int(g["pver"]), int(g["seed"]))
    d = eng.dev
    e = to_np(eng.zm_conv_evap(*[d(x) for x in ea[:7]], ea[7], d(ea[8].copy())))
    for k, v in e.items():
        assert_close("zm_conv_evap." + k, v, g["evap_" + k])


@pytest.mark.parametrize("pver,ncol,top_lev", [(32, 700, 1), (32, 700, 4), (128, 150, 1), (4, 90, 2)])
def test_modal_aero_vs_c_oracle(pver, ncol, top_lev):
    """top_lev > 1 (clim_modal_aero_top_lev, modal_aer_opt.F90:731, 1345): levels above it keep the initial zeros; the largest
    and the smallest level count the ABI accepts"""
    from c_oracle import COracle
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    props = synth.make_aero_props(5)
    props["top_lev"] = top_lev
    st = synth.make_state(ncol, pver, 5)
    aer = synth.make_aero_state(ncol, pver, props, 5)
    eng.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    d = eng.dev
    a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
    sw, sw_ref = to_np(eng.modal_aero_sw(*a)), co.modal_aero_sw(st, aer)
    lw, lw_ref = to_np(eng.modal_aero_lw(*a)), co.modal_aero_lw(st, aer)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, sw[k], sw_ref[k])
    assert_close("modal_aero_lw.tauxar", lw["tauxar"], lw_ref["tauxar"])
    if top_lev > 1:
        assert not sw["tauxar"][:, :, 1:top_lev].any() and not lw["tauxar"][:, :, :top_lev - 1].any()
        assert sw["tauxar"][:, :, top_lev:].all()


@pytest.mark.parametrize("family", ["bands-outer", "grouped"])
@pytest.mark.parametrize("nspec_m", [(5,), (1,), (2, 6), (8, 1, 3), (1, 1, 1, 1), (3, 2, 4, 1, 5), (2,) * 9])
def test_modal_aero_mode_and_species_counts_vs_c_oracle(nspec_m, family):
    """one mode, one species, the four-mode limit of the bands-outer kernels and what lies beyond it (five and nine modes take
    the grouped kernels whatever is asked for), up to eight species per mode: radiative outputs and history diagnostics"""
    from c_oracle import COracle
    pver, ncol = 32, 333
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    props = synth.make_aero_props(31, nspec_m=nspec_m)
    st = synth.make_state(ncol, pver, 31)
    aer = synth.make_aero_state(ncol, pver, props, 31)
    eng.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    eng.aero_kernel_select(family, family)
    try:
        d = eng.dev
        a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
        sw, sw_ref = to_np(eng.modal_aero_sw(*a)), co.modal_aero_sw(st, aer)
        lw, lw_ref = to_np(eng.modal_aero_lw(*a)), co.modal_aero_lw(st, aer)
        for k in ("tauxar", "wa", "ga", "fa"):
            assert_close("modal_aero_sw." + k, sw[k], sw_ref[k])
        assert_close("modal_aero_lw.tauxar", lw["tauxar"], lw_ref["tauxar"])
        assert not sw["status"].any() and not lw["status"].any()
        want = co.modal_aero_sw_diag(st, aer)
        for hand_over in (True, False):
            _assert_swd(_swd_engine(eng, st, aer, hand_over), want["lev"], want["col"], len(nspec_m))
    finally:
        eng.aero_kernel_select("auto", "auto")


@pytest.mark.parametrize("hand_over", [True, False])
@pytest.mark.parametrize("pver,ncol,top_lev", [(32, 701, 1), (32, 203, 4), (70, 100, 1), (26, 37, 2), (128, 130, 1), (4, 77, 1)])
def test_modal_aero_sw_diagnostics_vs_c_oracle(pver, ncol, top_lev, hand_over):
    """ragged column counts (blocks own NT/pver columns: 8, 3, 9 per block here), top_lev > 1, tropopause anywhere
    including above the top level and below the surface"""
    from c_oracle import COracle
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    props = synth.make_aero_props(12)
    props["top_lev"] = top_lev
    st = synth.make_state(ncol, pver, 12)
    aer = synth.make_aero_state(ncol, pver, props, 12)
    aer["troplevchem"] = aer["troplevchem"].astype(np.int32)
    aer["troplevchem"][::5] = 0
    aer["troplevchem"][1::7] = pver + 3
    aer["qaerwat"][:, ::9, :] *= -1.0                             # negative water: watervol reset before the partition
    eng.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    got, want = _swd_engine(eng, st, aer, hand_over), co.modal_aero_sw_diag(st, aer)
    _assert_swd(got, want["lev"], want["col"], eng.tables.nmodes)
    if top_lev > 1:
        assert not got["lev"][:, :, :top_lev - 1].any()


def test_modal_aero_full_size_properties():
    """ne30 size (48 602 columns x L32), no oracle needed: (1) cells are independent -- permuting the columns permutes every
    optics output bit for bit; (2) the optical depths are linear in the layer mass -- doubling pdeldry doubles tau, tau*w,
    tau*w*g, tau*w*g*g and the LW tau exactly (a power of two commutes with every rounding on the way)."""
    pver, ncol = 32, 48602
    eng = engine(cases.default_params(pver))
    props = synth.make_aero_props(4)
    st = synth.make_state(ncol, pver, 4)
    aer = synth.make_aero_state(ncol, pver, props, 4)
    eng.aero_tables_set(props, synth.PHYSCONST)
    d = eng.dev

    def run(pdeldry, mmr, dgn, qw):
        a = (d(pdeldry), d(mmr), d(dgn), d(qw))
        sw = to_np(eng.modal_aero_sw(*a)); lw = to_np(eng.modal_aero_lw(*a))
        return sw, lw

    sw, lw = run(st["pdeldry"], aer["specmmr"], aer["dgnumwet"], aer["qaerwat"])
    perm = np.random.default_rng(8).permutation(ncol)
    swp, lwp = run(st["pdeldry"][perm], aer["specmmr"][:, :, perm], aer["dgnumwet"][:, perm], aer["qaerwat"][:, perm])
    for k in ("tauxar", "wa", "ga", "fa"):
        assert np.array_equal(swp[k], sw[k][perm]), k
    assert np.array_equal(lwp["tauxar"], lw["tauxar"][perm])
    sw2, lw2 = run(2.0 * st["pdeldry"], aer["specmmr"], aer["dgnumwet"], aer["qaerwat"])
    for k in ("tauxar", "wa", "ga", "fa"):
        want = 2.0 * sw[k]
        want[:, :, 0] = sw[k][:, :, 0]                     # the zero'th layer holds presets, not aerosol (:639-642)
        assert np.array_equal(sw2[k], want), k
    assert np.array_equal(lw2["tauxar"], 2.0 * lw["tauxar"])


def test_modal_aero_sw_diagnostics_identities_at_full_size():
    """ne30 size (48 602 columns): relations that hold whatever the inputs -- the column sums are the level sums of the level
    fields times mass / air density, the species optical depths add up to the visible one, stratospheric parts are parts"""
    pver, ncol = 32, 48602
    eng = engine(cases.default_params(pver))
    props = synth.make_aero_props(3)
    st = synth.make_state(ncol, pver, 3)
    aer = synth.make_aero_state(ncol, pver, props, 3)
    eng.aero_tables_set(props, synth.PHYSCONST)
    dg = _swd_engine(eng, st, aer)
    lev, col = dg["lev"], dg["col"]
    mass = st["pdeldry"] * synth.PHYSCONST["rga"]
    ad = st["pmid"] / (synth.PHYSCONST["rair"] * st["t"])
    assert_close("AODVIS = sum extinct*mass/rho", (lev[0] * mass / ad).sum(1), col[0], rtol=1e-12)
    assert_close("AODABS = sum absorb*mass/rho", (lev[1] * mass / ad).sum(1), col[1], rtol=1e-12)
    assert_close("AODxASYM = sum asymext*mass/rho", (lev[2] * mass / ad).sum(1), col[5], rtol=1e-12)
    assert_close("AODUV", (lev[3] * mass / ad).sum(1), col[6], rtol=1e-12)
    assert_close("AODNIR", (lev[4] * mass / ad).sum(1), col[7], rtol=1e-12)
    assert_close("species AODs add up to AODVIS", col[17:23].sum(0), col[0], rtol=1e-10)
    nm = eng.tables.nmodes
    assert_close("mode AODs add up to AODVIS", col[24:24 + 3 * nm:3].sum(0), col[0], rtol=1e-12)
    assert_close("mode burdens add up to the species burdens", col[23:23 + 3 * nm:3].sum(0), col[10:16].sum(0), rtol=1e-12)
    for part, whole in ((3, 0), (8, 6), (9, 7)):
        assert (col[part] <= col[whole] * (1 + 1e-14)).all() and (col[part] >= 0).all()
    assert (col[2] >= col[0]).all()                               # all bands >= the visible band
    assert (col[4] <= col[0] * (1 + 1e-14)).all()                 # sum dopaer*palb <= sum dopaer


def _table_interp_intervals(g, x):
    """table_interp_calcwghts (oracle/stubs/l0_arithmetic.F90, restated in oracle/colphys_oracle_aero.c calcwghts): first grid
    point above x, extrapolation below the grid, clamp above it"""
    n = len(g)
    ii = np.searchsorted(g, x, side="right") + 1                 # first 1-based j with x < g_j
    ix1 = np.maximum(ii - 1, 1); ix2 = np.minimum(ix1 + 1, n)
    dx = g[ix2 - 1] - g[ix1 - 1]
    with np.errstate(invalid="ignore", divide="ignore"):
        t = np.where(np.abs(dx) > 1.0e-20, (x - g[ix1 - 1]) / np.where(dx == 0, 1.0, dx), 0.0)
    return ix1.astype(np.int32), ix2.astype(np.int32), t


@pytest.mark.parametrize("sw,mode,band", [(True, 0, 0), (True, 3, 13), (False, 1, 7)])
def test_table_interp_interval_search_scan_vs_lookup(sw, mode, band):
    """The optics kernels find table_interp's interval through a bucket lookup cut from the high word of x
    (colphys_common.h AERO_LUTB).  Walk every grid point, its floating-point neighbours, the ends, zero, negative, denormal,
    huge and infinite arguments through the lookup and through the reference's scan: same interval, bit-identical weight."""
    eng = engine(cases.default_params(32))
    props = synth.make_aero_props(17)
    eng.aero_tables_set(props, synth.PHYSCONST)
    rng = np.random.default_rng(99)
    for which, key in ((0, "refrtab"), (1, "refitab")):
        g = np.asarray(props[key + ("sw" if sw else "lw")][mode, band], dtype=np.float64)
        xs = [g]
        for steps in (1, 2, 3):
            up, dn = g.copy(), g.copy()
            for _ in range(steps):
                up, dn = np.nextafter(up, np.inf), np.nextafter(dn, -np.inf)
            xs += [up, dn]
        xs += [0.5 * (g[1:] + g[:-1]), np.array([0.0, -0.0, -1.0, -1.0e-300, 5.0e-324, 1.0e-310, 1.0e-30, 1.0e300, np.inf]),
               rng.uniform(-0.2, 3.0, 4000), 10.0 ** rng.uniform(-12, 2, 4000), g[-1] * (1 + 10.0 ** rng.uniform(-16, 0, 500)),
               g[1] * (1 - 10.0 ** rng.uniform(-16, 0, 500))]
        x = np.concatenate(xs)
        a1, a2, at, avail = eng.aero_grid_probe(sw, mode, band, which, False, eng.dev(x))
        b1, b2, bt, _ = eng.aero_grid_probe(sw, mode, band, which, True, eng.dev(x))
        assert avail, "the synthetic tables (linear real grid, geometric imaginary grid from 0) must admit the lookup"
        a1, a2, at, b1, b2, bt = (v.cpu().numpy() for v in (a1, a2, at, b1, b2, bt))
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2), np.argwhere(a1 != b1)[:5]
        assert np.array_equal(at, bt, equal_nan=True)                # x = inf: inf * 0 in both
        r1, r2, rt = _table_interp_intervals(g, x)
        assert np.array_equal(a1, r1) and np.array_equal(a2, r2)
        fin = np.isfinite(x)
        assert np.allclose(at[fin], rt[fin], rtol=1e-14, atol=1e-300)


def test_table_interp_lookup_on_random_grids():
    """The lookup's host-side construction over many grid shapes -- linear, geometric over 1 to 12 decades, starting at 0 or
    not, clustered -- either yields the scan's intervals exactly or is declined (then the kernels scan)."""
    eng = engine(cases.default_params(32))
    props = synth.make_aero_props(33)
    rng = np.random.default_rng(1234)
    made = declined = 0
    for trial in range(40):
        p = dict(props)
        kind = trial % 5
        def grid(n):
            if kind == 0:
                g = np.sort(rng.uniform(1.0, 3.0, n))
            elif kind == 1:
                g = np.geomspace(10.0 ** rng.uniform(-12, -2), 10.0 ** rng.uniform(-1, 1), n)
            elif kind == 2:
                g = np.concatenate([[0.0], np.geomspace(10.0 ** rng.uniform(-9, -3), rng.uniform(0.3, 2.0), n - 1)])
            elif kind == 3:
                g = np.cumsum(10.0 ** rng.uniform(-6, 0, n))                      # wildly uneven steps
            else:
                g = np.sort(rng.uniform(1.3, 1.3 + 10.0 ** rng.uniform(-9, -1), n))   # clustered: often declined
            return g
        p["refrtabsw"] = np.tile(grid(synth.PREFR), (props["nmodes"], synth.NSWBANDS, 1))
        p["refitabsw"] = np.tile(grid(synth.PREFI), (props["nmodes"], synth.NSWBANDS, 1))
        if not (np.all(np.diff(p["refrtabsw"][0, 0]) > 0) and np.all(np.diff(p["refitabsw"][0, 0]) > 0)):
            continue
        eng.aero_tables_set(p, synth.PHYSCONST)
        for which, key, n in ((0, "refrtabsw", synth.PREFR), (1, "refitabsw", synth.PREFI)):
            g = np.asarray(p[key][0, 0], dtype=np.float64)
            x = np.concatenate([g, np.nextafter(g, np.inf), np.nextafter(g, -np.inf), 0.5 * (g[1:] + g[:-1]),
                                np.array([0.0, 1e-300, 1e300]), rng.uniform(0.0, 2.0 * g[-1], 500),
                                10.0 ** rng.uniform(np.log10(max(g[0], 1e-15)) - 1, np.log10(g[-1]) + 1, 500)])
            a1, a2, at, avail = eng.aero_grid_probe(True, 0, 0, which, False, eng.dev(x))
            if not avail:
                declined += 1
                break
            b1, b2, bt, _ = eng.aero_grid_probe(True, 0, 0, which, True, eng.dev(x))
            assert torch_equal(a1, b1) and torch_equal(a2, b2) and np.array_equal(at.cpu().numpy(), bt.cpu().numpy(), equal_nan=True), (trial, which)
            r1, r2, _ = _table_interp_intervals(g, x)
            assert np.array_equal(a1.cpu().numpy(), r1) and np.array_equal(a2.cpu().numpy(), r2), (trial, which)
        else:
            made += 1
    assert made >= 15 and declined >= 1, (made, declined)
    eng.aero_tables_set(props, synth.PHYSCONST)


def torch_equal(a, b):
    return bool((a == b).all().item())


def test_modal_aero_tables_without_lookup_fall_back_to_the_scan(monkeypatch):
    """two real-index grid points one part in 1e12 apart share every bucket: no lookup for the SW tables, the kernels scan
    the grid; the LW tables keep theirs.  Then COLPHYS_AERO_NO_LUT forces the scan everywhere: bit-identical results."""
    from c_oracle import COracle
    pver, ncol = 32, 300
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    props = synth.make_aero_props(21)
    st = synth.make_state(ncol, pver, 21)
    aer = synth.make_aero_state(ncol, pver, props, 21)
    d = eng.dev
    a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
    eng.aero_tables_set(props, synth.PHYSCONST)
    sw_lut, lw_lut = to_np(eng.modal_aero_sw(*a)), to_np(eng.modal_aero_lw(*a))
    assert eng.aero_grid_probe(True, 0, 0, 0, True, d(np.ones(1)))[3] and eng.aero_grid_probe(False, 0, 0, 0, True, d(np.ones(1)))[3]
    monkeypatch.setenv("COLPHYS_AERO_NO_LUT", "1")
    eng.aero_tables_set(props, synth.PHYSCONST)
    assert not eng.aero_grid_probe(True, 0, 0, 0, False, d(np.ones(1)))[3]
    sw_scan, lw_scan = to_np(eng.modal_aero_sw(*a)), to_np(eng.modal_aero_lw(*a))
    for k in ("tauxar", "wa", "ga", "fa"):
        assert np.array_equal(sw_lut[k], sw_scan[k]), k
    assert np.array_equal(lw_lut["tauxar"], lw_scan["tauxar"])
    monkeypatch.delenv("COLPHYS_AERO_NO_LUT")
    props["refrtabsw"] = props["refrtabsw"].copy()
    props["refrtabsw"][1, 5, 3] = props["refrtabsw"][1, 5, 2] * (1.0 + 1.0e-12)
    eng.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    assert not eng.aero_grid_probe(True, 0, 0, 0, False, d(np.ones(1)))[3]
    assert eng.aero_grid_probe(False, 0, 0, 0, False, d(np.ones(1)))[3]
    sw, sw_ref = to_np(eng.modal_aero_sw(*a)), co.modal_aero_sw(st, aer)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, sw[k], sw_ref[k])


def test_modal_aero_edge_branches_vs_c_oracle():
    """the optics' side branches: negative aerosol water (:841-844), particles beyond the size table in both directions
    (:1559-1561 clamp, :875 large-particle extinction), refractive indices outside the table grid (table_interp
    extrapolation below / clamp above), near-empty cells"""
    from c_oracle import COracle
    pver, ncol = 32, 400
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    props = synth.make_aero_props(8)
    st = synth.make_state(ncol, pver, 8)
    aer = synth.make_aero_state(ncol, pver, props, 8)
    rng = np.random.default_rng(8)
    qw, dg, mm = aer["qaerwat"], aer["dgnumwet"], aer["specmmr"]
    qw[:, ::7, :] *= -1.0                                        # negative water volume
    qw[:, 3::11, :] *= 1.0e3                                     # water-dominated: index pulled to water's, off the grid
    dg[:, 1::5, :] = 10.0 ** rng.uniform(-4.6, -3.5, dg[:, 1::5, :].shape)    # > 25 um surface-mode radius
    dg[:, 2::9, :] = 10.0 ** rng.uniform(-9.5, -8.2, dg[:, 2::9, :].shape)    # < 0.01 um
    mm[:, :, 4::13, :] *= 1.0e-8                                 # almost no dry aerosol
    eng.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    d = eng.dev
    a = (d(st["pdeldry"]), d(mm), d(dg), d(qw))
    sw, sw_ref = to_np(eng.modal_aero_sw(*a)), co.modal_aero_sw(st, aer)
    lw, lw_ref = to_np(eng.modal_aero_lw(*a)), co.modal_aero_lw(st, aer)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, sw[k], sw_ref[k])
    assert_close("modal_aero_lw.tauxar", lw["tauxar"], lw_ref["tauxar"])


# ---------------------------------------------------------------------------------------------------
# (c) properties at BASELINE.json's full size
# ---------------------------------------------------------------------------------------------------
def test_full_size_column_independence_and_gather(monkeypatch):
    """48 602 columns x L32: permuting the columns permutes the per-column results bit for bit; changing the
    chunk width changes only the gathered layout; the gathered layout is a faithful ascending compaction."""
    ncol, pver = 48602, 32
    eng = engine(cases.default_params(pver))
    st = synth.make_state(ncol, pver, 77)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    o16 = run_zm(eng, args, pcols=16)
    full = ("qtnd", "heat", "cme", "zdu", "rprd", "dlf", "ql", "eurt", "mcon", "pflx", "prec", "jctop", "jcbot", "cape", "rliq")
    for k in full:
        assert np.isfinite(o16[k]).all(), k
    deep = o16["jctop"] < pver            # deep columns have a plume top above the lowest level
    nd = int(o16["lengath"].sum())
    assert 0.03 * ncol < nd < 0.5 * ncol
    # permutation
    perm = np.random.default_rng(3).permutation(ncol)
    pargs = tuple(a[perm] if isinstance(a, np.ndarray) else a for a in args)
    op = run_zm(eng, pargs, pcols=16)
    for k in full:
        assert np.array_equal(op[k], o16[k][perm]), f"{k} depends on column order"
    # chunk width
    o48 = run_zm(eng, args, pcols=48)
    for k in full:
        assert np.array_equal(o48[k], o16[k]), f"{k} depends on pcols"
    # two columns per wave in the plume kernels (what 150 000 columns and more select): which column shares a wave with
    # which follows the order the trigger's atomics listed them in, and must not matter
    monkeypatch.setenv("COLPHYS_PLUME_CPW", "2")
    o2 = run_zm(eng, args, pcols=16)
    monkeypatch.delenv("COLPHYS_PLUME_CPW")
    for k in full + ("mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg", "ideep", "lengath"):
        assert np.array_equal(o2[k], o16[k]), f"{k} depends on the plume kernels' columns per wave"
    # gathered layout: ideep ascending per chunk, rows equal the full-layout values they came from
    for pcols, o in ((16, o16), (48, o48)):
        m = gathered_mask(o["lengath"], ncol, pcols)
        rows = np.flatnonzero(m)
        src = (rows // pcols) * pcols + o["ideep"][rows] - 1
        assert int(m.sum()) == nd
        assert np.all(np.diff(src) > 0)
        assert np.array_equal(o["jt"][rows], o["jctop"][src].astype(np.int32))
        assert np.array_equal(o["maxg"][rows], o["jcbot"][src].astype(np.int32))
        assert np.array_equal(o["du"][rows], o["zdu"][src])          # zdu is the scattered du (zm_conv.F90:1504)
        np.testing.assert_allclose(o["mu"][rows] + o["md"][rows], o["mcon"][src][:, :pver], rtol=1e-12, atol=1e-18)  # mc = mu + md
        assert np.all(o["mu"][rows] >= 0) and np.all(o["md"][rows] <= 0)
        assert np.all((o["prec"] >= 0)) and np.all(o["prec"][~np.isin(np.arange(ncol), src)] == 0)


def test_ne120_size_repeats_its_tiles():
    """777 728 columns (BASELINE.json's ne120-sized config, rounded to whole chunks of a 48 608-column slab repeated 16 times):
    every repetition gives the first one's results bit for bit -- per-column outputs and, the slab being whole chunks, the
    gathered sets too -- so no index or offset breaks beyond 2^31 bytes; the optics the same way at 388 864 columns, where the
    four SW outputs are 1.4 GB each."""
    base, rep, pver = 48608, 16, 32
    eng = engine(cases.default_params(pver))
    st = synth.make_state(base, pver, 120)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    big = tuple(np.tile(a, (rep,) + (1,) * (a.ndim - 1)) if isinstance(a, np.ndarray) else a for a in args)
    o = run_zm(eng, big)
    assert o["lengath"].sum() > 0.03 * base * rep and not o["status"].any()
    small = run_zm(eng, args)             # the slab alone takes other kernel variants (wave-pair trigger, one column per plume wave)
    nch = base // 16
    for k, v in o.items():
        n0 = nch if k == "lengath" else base
        first = v[:n0]
        assert np.array_equal(first, small[k]), k
        for r in (1, rep // 2, rep - 1):
            assert np.array_equal(v[r * n0:(r + 1) * n0], first), (k, r)
    del o, big
    props = synth.make_aero_props(120)
    aer = synth.make_aero_state(base, pver, props, 120)
    eng.aero_tables_set(props, synth.PHYSCONST)
    rep = 8
    d = eng.dev
    tile = lambda a, ax: np.concatenate([a] * rep, axis=ax)
    a = (d(tile(st["pdeldry"], 0)), d(tile(aer["specmmr"], 2)), d(tile(aer["dgnumwet"], 1)), d(tile(aer["qaerwat"], 1)))
    sw = eng.modal_aero_sw(*a)
    lw = eng.modal_aero_lw(*a)
    eng.torch.cuda.synchronize()
    for name, v in (("tauxar", sw["tauxar"]), ("wa", sw["wa"]), ("ga", sw["ga"]), ("fa", sw["fa"]), ("lw", lw["tauxar"])):
        first = v[:base]
        assert bool((first > 0).any()), name
        for r in (1, rep - 1):
            assert bool(eng.torch.equal(v[r * base:(r + 1) * base], first)), (name, r)
    assert not bool(sw["status"].any()) and not bool(lw["status"].any())
    del sw, lw, a
    eng.torch.cuda.empty_cache()


@pytest.mark.parametrize("pver", [32, 70])
def test_full_size_water_budgets(pver):
    """48 602 columns x L32 and x L70 (BASELINE.json's high-top config), the closed budgets the reference's formulation implies:
    zm_convr -- precipitation is the column-integrated drying minus the detrained liquid (zm_conv.F90:1629-1638);
    zm_conv_evap -- the precipitation flux at the surface is the column sum of the net production, what evaporates is
    what moistens, and the heating is the latent heat of those two (zm_conv.F90:1943-1966)."""
    ncol = 48602
    eng = engine(cases.default_params(pver))
    st = synth.make_state(ncol, pver, 78)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    o = run_zm(eng, args)
    g = synth.PHYSCONST["gravit"]
    col = -((o["qtnd"] + o["dlf"]) * st["pdel"]).sum(axis=1) / g / 1000.0
    deep = o["prec"] > 0
    assert deep.sum() > 0.03 * ncol
    np.testing.assert_allclose(o["prec"][deep], col[deep], rtol=1e-9)
    assert not col[~deep].any()
    d = eng.dev
    q1 = st["q"] + synth.ZTODT * o["qtnd"]                      # the state zm_conv_evap sees (zm_conv_intr.F90:739-764)
    t1 = st["t"] + synth.ZTODT * o["heat"] / synth.PHYSCONST["cpair"]
    e = to_np(eng.zm_conv_evap(d(t1), d(st["pmid"]), d(st["pdel"]), d(q1), d(st["landfrac"]), d(o["rprd"]), d(st["cld"]),
                               synth.ZTODT, d(o["prec"].copy())))
    for k, v in e.items():
        assert np.isfinite(v).all(), k
    net = (e["ntprprd"] * st["pdel"]).sum(axis=1) / g
    np.testing.assert_allclose(e["flxprec"][:, pver], net, rtol=1e-9, atol=1e-14)    # up to the clamps at zero (:1947)
    assert np.array_equal(e["prec"], e["flxprec"][:, pver] / 1000.0)
    assert np.array_equal(e["snow"], e["flxsnow"][:, pver] / 1000.0)
    assert (e["flxprec"] >= 0).all() and (e["flxsnow"] >= 0).all() and (e["tend_q"] >= 0).all()
    evp = o["rprd"] - e["ntprprd"]                               # evaporated precipitation = the moistening
    np.testing.assert_allclose(e["tend_q"], evp, rtol=1e-9, atol=1e-16)
    assert (e["tend_q"] > 0).sum() > 1000


@pytest.mark.parametrize("staging", ["batched", "per-field"])
def test_chunk_entry_points_match_device_entry_points(staging, monkeypatch):
    """The Fortran-facing (pcols,pver,nchunks) host-layout calls give the same numbers as the device calls -- with the
    staging of a few-column call (inputs up / outputs down in one copy each) and with the per-field pipeline of a many-chunk
    call (forced here at the same small size)."""
    import ctypes as C
    monkeypatch.setenv("COLPHYS_STAGE_BATCH", "1" if staging == "batched" else "0")
    pver, pcols = 32, 16
    ncols = np.array([16, 16, 9, 16, 1], dtype=np.int32)       # ragged chunks, as ne30's last chunk
    nch = len(ncols)
    eng = engine(cases.default_params(pver))
    st = synth.make_state(int(ncols.sum()), pver, 31)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    # flat reference run: every chunk padded to pcols by repeating its first column is NOT what the flat call
    # does, so compare per chunk against a flat run of exactly that chunk's columns
    offs = np.concatenate([[0], np.cumsum(ncols)])

    def to_chunks(a, nlev):
        out = np.full((nch, nlev, pcols), np.inf)      # Fortran (pcols, nlev, nchunks)
        for c in range(nch):
            blk = a[offs[c]:offs[c + 1]].reshape(ncols[c], nlev)
            out[c, :, :ncols[c]] = blk.T
        return np.ascontiguousarray(out)

    names_mid = ("t", "qh", "pap", "dpp", "zm")
    vals = dict(t=args[0], qh=args[1], pblh=args[2], zm=args[3], geos=args[4], zi=args[5], pap=args[6], paph=args[7],
                dpp=args[8], tpert=args[10], landfrac=args[11])
    hin = {k: to_chunks(vals[k], pver) for k in names_mid}
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
    rc = eng.lib.colphys_zm_convr_chunks(C.c_int(nch), C.c_int(pcols), C.c_int(pver), abi.iptr(ncols), C.c_double(args[9]), C.byref(io))
    assert rc == 0, eng.lib.colphys_last_error()
    for c in range(nch):
        sl = slice(offs[c], offs[c + 1])
        cargs = tuple(a[sl] if isinstance(a, np.ndarray) else a for a in args)
        o = run_zm(eng, cargs, pcols=pcols)
        n = ncols[c]
        assert hout["lengath"][c] == o["lengath"][0]
        lg = int(o["lengath"][0])
        for k in ("qtnd", "heat", "rprd", "mcon", "pflx", "dlf", "ql"):
            assert np.array_equal(hout[k][c, :, :n].T, o[k]), k
        for k in ("prec", "cape", "jctop", "rliq"):
            assert np.array_equal(hout[k][c, 0, :n], o[k]), k
        for k in ("mu", "md", "eu", "dp"):
            assert np.array_equal(hout[k][c, :, :lg].T, o[k][:lg]), k
        for k in ("jt", "maxg", "ideep"):
            assert np.array_equal(hout[k][c, 0, :lg], o[k][:lg]), k


def test_chunk_entry_is_safe_to_call_from_a_threaded_chunk_loop():
    """The reference calls the path from inside an OpenMP loop over chunks (physpkg.F90:1147): concurrent per-chunk calls
    from several host threads (ctypes drops the GIL during the call) give what the same calls give one after another."""
    import ctypes as C
    import threading
    pver, pcols, nthr, rounds = 32, 16, 4, 3
    eng = engine(cases.default_params(pver))

    def make_call(seed, ncol):
        st = synth.make_state(ncol, pver, seed)
        a = cases.zm_args_from_state(st, st["zi"], st["zm"])
        vals = dict(t=a[0], qh=a[1], pblh=a[2], zm=a[3], geos=a[4], zi=a[5], pap=a[6], paph=a[7], dpp=a[8], tpert=a[10], landfrac=a[11])

        def chunk(x, nlev):
            out = np.full((1, nlev, pcols), np.inf)
            out[0, :, :ncol] = x.reshape(ncol, nlev).T
            return out
        hin = {k: chunk(vals[k], pver) for k in ("t", "qh", "pap", "dpp", "zm")}
        hin.update({k: chunk(vals[k], pver + 1) for k in ("paph", "zi")})
        hin.update({k: chunk(vals[k], 1) for k in ("geos", "pblh", "tpert", "landfrac")})

        def run():
            hout = {k: np.zeros((1, pver, pcols)) for k in abi.ZM_CONVR_OUT_MID}
            hout.update({k: np.zeros((1, pver + 1, pcols)) for k in abi.ZM_CONVR_OUT_INT})
            hout.update({k: np.zeros((1, 1, pcols)) for k in abi.ZM_CONVR_OUT_COL})
            hout.update({k: np.zeros((1, 1, pcols), dtype=np.int32) for k in abi.ZM_CONVR_OUT_ICOL})
            hout["lengath"] = np.zeros(1, dtype=np.int32)
            io = abi.ZmConvrIO()
            for k, v in {**hin, **hout}.items():
                setattr(io, k, abi.iptr(v) if v.dtype == np.int32 else abi.dptr(v))
            nc = np.array([ncol], dtype=np.int32)
            rc = eng.lib.colphys_zm_convr_chunks(C.c_int(1), C.c_int(pcols), C.c_int(pver), abi.iptr(nc), C.c_double(a[9]), C.byref(io))
            # the next call of the same thread's chunk iteration: another entry point through the same staging
            ein = dict(t=hin["t"], pmid=hin["pap"], pdel=hin["dpp"], q=hin["qh"], prdprec=hout["rprd"], cldfrc=ecld,
                       landfrac=hin["landfrac"], prec=hout["prec"].copy())
            eout = {k: np.zeros((1, pver, pcols)) for k in ("tend_s", "tend_s_snwprd", "tend_s_snwevmlt", "tend_q", "ntprprd", "ntsnprd")}
            eout.update({k: np.zeros((1, pver + 1, pcols)) for k in ("flxprec", "flxsnow")})
            eout["snow"] = np.zeros((1, 1, pcols))
            eio = abi.ZmConvEvapIO()
            for k, v in {**ein, **eout}.items():
                setattr(eio, k, abi.dptr(v))
            rc2 = eng.lib.colphys_zm_conv_evap_chunks(C.c_int(1), C.c_int(pcols), C.c_int(pver), abi.iptr(nc), C.c_double(synth.ZTODT),
                                                      C.byref(eio))
            hout.update({"evap_" + k: v for k, v in eout.items()})
            hout["evap_prec"] = ein["prec"]
            return rc or rc2, hout
        ecld = chunk(st["cld"], pver)
        return run

    calls = [make_call(500 + i, (16, 9, 16, 1)[i % 4]) for i in range(nthr)]
    want = []
    for run in calls:
        rc, h = run()
        assert rc == 0, eng.lib.colphys_last_error()
        want.append(h)
    assert sum(int(h["lengath"][0]) for h in want) > 0 and any(h["evap_tend_q"].any() for h in want)
    got = [[None] * rounds for _ in range(nthr)]

    def worker(i):
        for r in range(rounds):
            got[i][r] = calls[i]()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(nthr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive()
    for i in range(nthr):
        for r in range(rounds):
            rc, h = got[i][r]
            assert rc == 0
            for k, v in h.items():
                assert np.array_equal(v, want[i][k]), (i, r, k)


def test_endrun_conditions_are_reported_per_column():
    """What the reference turns into endrun() comes back as per-column status bits, for exactly the columns the C restatement
    flags, and leaves every other column's results untouched: a sounding that is not a number ends Brent's iteration
    unconverged (zm_conv.F90:5401-5410, the `ientropy` abort a model blow-up ends in), an aerosol load beyond the sanity range
    of the layer optical depth (modal_aer_opt.F90:997-1029 SW: 30; :1402-1437 LW: 20)."""
    from c_oracle import COracle
    pver, ncol = 32, 300
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    st = synth.make_state(ncol, pver, 5)
    args = list(cases.zm_args_from_state(st, st["zi"], st["zm"]))
    clean = run_zm(eng, tuple(args))
    assert not clean["status"].any()
    t, q = args[0].copy(), args[1].copy()
    t[7, :] = np.nan; t[50, 20] = np.nan; t[90, pver - 1] = np.nan; q[120, 25] = np.nan; t[299, :] = np.nan
    bad = np.array([7, 50, 90, 120, 299])
    args[0], args[1] = t, q
    ref = co.zm_convr(*args)
    o = run_zm(eng, tuple(args))
    assert np.array_equal(o["status"], ref["status"])
    # a sounding without a number anywhere fails in the first inversion the ascent makes, an ienthalpy (zm_conv.F90:5102)
    assert (o["status"][[7, 299]] == abi.STATUS_BRENT_NOT_CONVERGED | abi.STATUS_BRENT_ENTHALPY).all()
    ok = np.setdiff1d(np.arange(ncol), bad)
    okc = np.setdiff1d(np.arange(ncol), np.concatenate([np.arange(b // 16 * 16, b // 16 * 16 + 16) for b in bad]))
    for k in abi.ZM_CONVR_OUT_MID + abi.ZM_CONVR_OUT_INT + abi.ZM_CONVR_OUT_COL:
        rows = okc if k in abi.ZM_CONVR_GATHERED else ok       # gathered rows of a chunk shift when a neighbour drops out
        assert np.array_equal(o[k][rows], clean[k][rows]), k
    # optics
    props = synth.make_aero_props(9)
    aer = synth.make_aero_state(ncol, pver, props, 9)
    heavy = np.array([3, 64, 65, 298])
    aer["specmmr"][:, :, heavy, 10:14] *= 3.0e6                 # specmmr [mode][species][col][lev]
    pdeldry = st["pdeldry"].copy()
    pdeldry[17, 5] *= -1.0e4                                    # a negative layer mass: negative optical depths, SW and LW
    st = dict(st, pdeldry=pdeldry)
    eng.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    d = eng.dev
    a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
    sw, lw = to_np(eng.modal_aero_sw(*a)), to_np(eng.modal_aero_lw(*a))
    sw_ref, lw_ref = co.modal_aero_sw(st, aer), co.modal_aero_lw(st, aer)
    assert np.array_equal(sw["status"], sw_ref["status"]) and np.array_equal(lw["status"], lw_ref["status"])
    bits, count = sw["status"] & 0xff, sw["status"] >> abi.STATUS_COUNT_SHIFT
    assert set(np.flatnonzero(bits)) == set(heavy) | {17}
    assert (bits[heavy] == abi.STATUS_DOPAER_RANGE).all()                                    # large: the reference warns and counts
    assert bits[17] == abi.STATUS_DOPAER_RANGE | abi.STATUS_DOPAER_NEGATIVE                  # negative: it halts
    assert (count[heavy] >= 4).all() and count[17] == 4 * synth.NSWBANDS and (count[bits == 0] == 0).all()
    lbits = lw["status"] & 0xff
    assert lbits[17] == abi.STATUS_DOPAER_RANGE | abi.STATUS_DOPAER_NEGATIVE and (lbits[heavy] == abi.STATUS_DOPAER_RANGE).all()
    assert_close("modal_aero_sw.tauxar", sw["tauxar"], sw_ref["tauxar"])


def test_argument_errors_are_refused_before_any_launch():
    """zm_convi's aborts and the argument checks of the C ABI: an unsupported option or a malformed call returns its error code
    with a message and leaves the engine as it was (the next correct call gives the same numbers as before)."""
    import copy
    import ctypes as C
    from cam_nor_physics_amd.engine import ColphysError
    pver, ncol = 32, 64
    params = cases.default_params(pver)
    eng = engine(params)
    st = synth.make_state(ncol, pver, 21)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    before = run_zm(eng, args)
    lib = eng.lib
    for field, code in (("zmconv_microp", -5), ("zm_org", -5), ("cam3", -5)):      # COLPHYS_ERR_UNSUPPORTED (zm_conv.F90:3782, :75-78)
        bad = copy.copy(params)
        setattr(bad, field, 1)
        assert lib.colphys_init(0, C.byref(bad)) == code, field
        assert lib.colphys_last_error()
    assert lib.colphys_init(99, C.byref(params)) == -1                              # no such device
    bad = copy.copy(params)
    bad.limcnv = pver                                                               # no level left under the convection top
    assert lib.colphys_init(0, C.byref(bad)) == 0
    with pytest.raises(ColphysError, match="limcnv"):
        run_zm(eng, args)
    assert lib.colphys_init(0, C.byref(params)) == 0
    d = eng.dev
    o = eng.alloc_zm_convr_out(ncol, pver, 16)
    io = abi.ZmConvrIO()                                                            # every pointer null
    assert lib.colphys_zm_convr_dev(C.c_int(ncol), C.c_int(pver), C.c_int(16), C.c_double(900.0), C.byref(io), None) == -1
    assert b"null pointer" in lib.colphys_last_error()
    assert lib.colphys_zm_convr_dev(C.c_int(ncol), C.c_int(pver), C.c_int(16), C.c_double(900.0), None, None) == -1
    for n_, pv_, pc_ in ((0, pver, 16), (-3, pver, 16), (ncol, 3, 16), (ncol, 129, 16), (ncol, pver, 0), (ncol, pver, 5000)):
        assert lib.colphys_zm_convr_dev(C.c_int(n_), C.c_int(pv_), C.c_int(pc_), C.c_double(900.0), C.byref(io), None) == -1, (n_, pv_, pc_)
    nc = np.array([16, 17], dtype=np.int32)                                         # more columns than the chunk is wide
    assert lib.colphys_zm_convr_chunks(C.c_int(2), C.c_int(16), C.c_int(pver), abi.iptr(nc), C.c_double(900.0), C.byref(io)) == -1
    nc[1] = 0
    assert lib.colphys_zm_convr_chunks(C.c_int(2), C.c_int(16), C.c_int(pver), abi.iptr(nc), C.c_double(900.0), C.byref(io)) == -1
    assert lib.colphys_aero_kernel_select(7, 0) == -1
    assert lib.colphys_profile_get(99, None, None) != 0
    after = run_zm(eng, args)
    for k in before:
        assert np.array_equal(before[k], after[k]), k
    del d, o


def test_finalize_and_reinitialise():
    """colphys_finalize releases everything the library holds: compute calls are refused until colphys_init, the optics until
    their tables are set again, and the second life gives the numbers of the first (three cycles: nothing accumulates)."""
    import ctypes as C
    from cam_nor_physics_amd.engine import ColphysError
    pver, ncol = 32, 200
    params = cases.default_params(pver)
    eng = engine(params)
    props = synth.make_aero_props(3)
    st = synth.make_state(ncol, pver, 3)
    aer = synth.make_aero_state(ncol, pver, props, 3)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])

    def life():
        eng.aero_tables_set(props, synth.PHYSCONST)
        d = eng.dev
        z = run_zm(eng, args)
        lw = to_np(eng.modal_aero_lw(d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"])))
        return z, lw
    z0, lw0 = life()
    for _ in range(3):
        assert eng.lib.colphys_finalize() == 0 and eng.lib.colphys_is_initialized() == 0
        with pytest.raises(ColphysError, match="colphys_init"):
            run_zm(eng, args)
        assert eng.lib.colphys_finalize() == 0                     # twice is harmless
        assert eng.lib.colphys_init(0, C.byref(params)) == 0 and eng.lib.colphys_is_initialized() == 1
        d = eng.dev
        with pytest.raises(ColphysError, match="colphys_aero_tables_set"):
            eng.modal_aero_lw(d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
        z, lw = life()
        for k in z0:
            assert np.array_equal(z[k], z0[k]), k
        assert np.array_equal(lw["tauxar"], lw0["tauxar"])


def test_no_column_convects():
    """dry soundings: every chunk takes the reference's early return (lengath = 0, zm_conv.F90:917), outputs are the
    initial zeros / defaults, and the second-ascent / plume launches find an empty list"""
    from c_oracle import COracle
    pver, ncol = 32, 700
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    st = synth.make_state(ncol, pver, 91, rh_range=(0.05, 0.3))
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    ref = co.zm_convr(*args)
    assert ref["lengath"].sum() == 0
    o = run_zm(eng, args)
    check_zm(o, ref, ncol)
    assert not o["heat"].any() and not o["prec"].any() and (o["jctop"] == pver).all() and (o["jcbot"] == 1).all()


@pytest.mark.parametrize("ncol", [1, 5, 63, 64, 65, 257, 300])
def test_ragged_column_counts_vs_c_oracle(ncol):
    """block tails of every kernel on the path: partial waves, partial column groups of the trigger's wave pairs, a
    partial last chunk, and the column-count-one case"""
    from c_oracle import COracle
    pver = 32
    params = cases.default_params(pver)
    eng = engine(params)
    co = COracle(params)
    st = synth.make_state(ncol, pver, 77 + ncol, rh_range=(0.75, 0.98))          # moist: most columns convect
    c = synth.PHYSCONST
    d = eng.dev
    ga = cases.geo_args(st)
    zi_r, zm_r = co.geopotential_t(*ga, st["q"], st["rair"], c["gravit"], st["zvir"], dycore=abi.DYCORE_FV)
    zi, zm = eng.geopotential_t(*[d(x) for x in ga], d(st["q"]), d(st["rair"]), c["gravit"], d(st["zvir"]))
    assert_close("geopotential_t.zi", zi.cpu().numpy(), zi_r)
    args = cases.zm_args_from_state(st, zi_r, zm_r)
    ref = co.zm_convr(*args)
    o = run_zm(eng, args)
    check_zm(o, ref, ncol)
    dt = synth.ZTODT
    ea = (st["t"], st["pmid"], st["pdel"], st["q"], st["landfrac"], ref["rprd"], st["cld"], dt, ref["prec"])
    e_ref = co.zm_conv_evap(*ea)
    e = to_np(eng.zm_conv_evap(*[d(x) for x in ea[:7]], dt, d(ref["prec"].copy())))
    for k, v in e.items():
        assert_close("zm_conv_evap." + k, v, e_ref[k])
    m_ref = co.momtran(st["u"], st["v"], ref["mu"], ref["md"], ref["du"], ref["eu"], ref["ed"], ref["dp"], ref["dsubcld"],
                       ref["jt"], ref["maxg"], ref["ideep"], ref["lengath"], dt)
    m = to_np(eng.momtran(d(np.stack([st["u"], st["v"]])), *[d(ref[k]) for k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld")],
                          *[d(ref[k]) for k in ("jt", "maxg", "ideep", "lengath")], dt))
    for k, v in m.items():
        assert_close("momtran." + k, v, m_ref[k])
    props = synth.make_aero_props(3)
    aer = synth.make_aero_state(ncol, pver, props, 3)
    eng.aero_tables_set(props, c)
    co.aero_tables_set(props, c)
    a = (d(st["pdeldry"]), d(aer["specmmr"]), d(aer["dgnumwet"]), d(aer["qaerwat"]))
    sw, sw_ref = to_np(eng.modal_aero_sw(*a)), co.modal_aero_sw(st, aer)
    lw, lw_ref = to_np(eng.modal_aero_lw(*a)), co.modal_aero_lw(st, aer)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, sw[k], sw_ref[k])
    assert_close("modal_aero_lw.tauxar", lw["tauxar"], lw_ref["tauxar"])


@pytest.mark.parametrize("pver,ncol", [(32, 2000), (70, 300), (128, 200), (12, 700), (4, 100)])
def test_zm_conv_tend_resident_matches_composed_calls(pver, ncol):
    """colphys_zm_conv_tend_dev (zm_conv_intr.F90:441-905 on resident data) against the same sequence composed call by
    call -- zm_convr, update, geopotential_t, zm_conv_evap, update, geopotential_t, momtran, update, geopotential_t --
    and against numpy restatements of the host-side post-processing (:685-731) and of physics_ptend_sum."""
    eng = engine(cases.default_params(pver))
    st = synth.make_state(ncol, pver, 404)
    c = synth.PHYSCONST
    d = eng.dev
    dt, pcols = synth.ZTODT, 16
    S = {k: d(st[k]) for k in ("t", "q", "u", "v", "pmid", "pint", "pdel", "rpdel", "lnpint", "lnpmid", "zm", "zi", "phis",
                               "rair", "zvir", "pblh", "tpert", "landfrac", "cld")}
    S["ps"] = d(st["pint"][:, -1].copy())
    o = to_np(eng.zm_conv_tend(S, dt, pcols=pcols))

    z = to_np(eng.zm_convr(S["t"], S["q"], S["pblh"], S["zm"], S["phis"], S["zi"], S["pmid"], S["pint"], S["pdel"], 0.5 * dt,
                           S["tpert"], S["landfrac"], pcols=pcols))
    qmin = 1.0e-12
    t1 = st["t"] + z["heat"] * dt / c["cpair"]
    q1 = np.maximum(st["q"] + z["qtnd"] * dt, qmin)
    prec = d(z["prec"].copy())
    e = to_np(eng.zm_conv_evap(d(t1), S["pmid"], S["pdel"], d(q1), S["landfrac"], d(z["rprd"]), S["cld"], dt, prec))
    t2 = t1 + e["tend_s"] * dt / c["cpair"]
    q2 = np.maximum(q1 + e["tend_q"] * dt, qmin)
    m = to_np(eng.momtran(d(np.stack([st["u"], st["v"]])), *[d(z[k]) for k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld")],
                          *[d(z[k]) for k in ("jt", "maxg", "ideep", "lengath")], dt, pcols=pcols))
    t3 = t2 + m["seten"] * dt / c["cpair"]
    u3 = st["u"] + m["dqdt"][0] * dt
    v3 = st["v"] + m["dqdt"][1] * dt
    zi3, zm3 = eng.geopotential_t(S["lnpint"], S["lnpmid"], S["pint"], S["pmid"], S["pdel"], S["rpdel"], d(t3), d(q2), S["rair"],
                                  c["gravit"], S["zvir"])

    for k in ("ql", "rprd", "dlf", "mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg", "ideep", "lengath", "cape",
              "rliq", "rice", "jctop", "jcbot"):
        assert np.array_equal(o[k], z[k]), k
    assert np.array_equal(o["mconzm"], z["mcon"] * 100.0 / c["gravit"])
    assert np.array_equal(o["prec"], e["prec"]) and np.array_equal(o["snow"], e["snow"])
    assert np.array_equal(o["evapcdp"], e["tend_q"])
    assert np.array_equal(o["flxprec"], e["flxprec"]) and np.array_equal(o["flxsnow"], e["flxsnow"])
    assert np.array_equal(o["ptend_s"], (z["heat"] + e["tend_s"]) + m["seten"])
    assert np.array_equal(o["ptend_q"], z["qtnd"] + e["tend_q"])
    assert np.array_equal(o["ptend_u"], m["dqdt"][0]) and np.array_equal(o["ptend_v"], m["dqdt"][1])
    assert np.array_equal(o["t1"], t3) and np.array_equal(o["q1"], q2)
    assert np.array_equal(o["u1"], u3) and np.array_equal(o["v1"], v3)
    assert np.array_equal(o["zi1"], zi3.cpu().numpy()) and np.array_equal(o["zm1"], zm3.cpu().numpy())
    # host-side post-processing of zm_conv_tend, restated
    freq = np.zeros(ncol); mu_out = np.zeros((ncol, pver)); md_out = np.zeros((ncol, pver))
    pcont = st["pint"][:, -1].copy(); pconb = pcont.copy()
    for ic, lg in enumerate(z["lengath"]):
        for j in range(lg):
            row = ic * pcols + j
            ii = ic * pcols + z["ideep"][row] - 1
            freq[ii] = 1.0
            mu_out[ii] = z["mu"][row] * 100.0 / c["gravit"]
            md_out[ii] = z["md"][row] * 100.0 / c["gravit"]
            if z["maxg"][row] > z["jt"][row]:
                pcont[ii] = st["pmid"][ii, z["jt"][row] - 1]
                pconb[ii] = st["pmid"][ii, z["maxg"][row] - 1]
    assert (z["lengath"].sum() > 0) == (pver >= 12)      # four levels: nothing convects, every kernel still runs
    for k, r in (("freqzm", freq), ("mu_out", mu_out), ("md_out", md_out), ("pcont", pcont), ("pconb", pconb)):
        assert np.array_equal(o[k], r), k


# ---------------------------------------------------------------------------------------------------
# (d) the device logarithms that stand where the reference calls its libm
# ---------------------------------------------------------------------------------------------------
def test_device_logarithms_within_one_ulp_class():
    """cp_log / cp_log10 (colphys_common.h) against numpy's libm over the ranges the kernels use and beyond:
    <= 2 ulp everywhere (both sides are < 1 ulp implementations), exceptional arguments as libm."""
    eng = engine(cases.default_params(32))
    rng = np.random.default_rng(11)
    x = np.concatenate([
        rng.uniform(0.5, 2.0, 400000),                      # T/Tfreez, tboil/T, interface pressure ratios
        10.0 ** rng.uniform(-12, 6, 400000),                # qv/qst, (p-e)/pref and far beyond
        np.nextafter(1.0, 0) ** rng.integers(1, 1000, 1000), 1.0 + rng.uniform(-1e-8, 1e-8, 100000),
        np.array([1.0, 2.0, 0.5, 4.9e-324, 2.2e-308, 1.7e308]),
    ])
    with np.errstate(all="ignore"):
        for kind, ref in (("log", np.log), ("log10", np.log10)):
            y = eng.math_probe(kind, eng.dev(x)).cpu().numpy()
            r = ref(x)
            ulp = np.abs(y - r) / np.spacing(np.abs(r))
            ulp[r == 0.0] = np.abs(y[r == 0.0]) / np.spacing(1.0) * 0 + (y[r == 0.0] != 0.0)
            assert ulp.max() <= 2.0, (kind, ulp.max(), x[np.argmax(ulp)])
            bad = np.array([0.0, -1.0, np.inf, np.nan])
            yb = eng.math_probe(kind, eng.dev(bad)).cpu().numpy()
            assert yb[0] == -np.inf and np.isnan(yb[1]) and yb[2] == np.inf and np.isnan(yb[3])
    t = rng.uniform(170.0, 330.0, 200000)
    es = eng.math_probe("svp_water", eng.dev(t)).cpu().numpy()
    tb = 373.16
    es_ref = 10.0 ** (-7.90298 * (tb / t - 1.0) + 5.02808 * np.log10(tb / t) - 1.3816e-7 * (10.0 ** (11.344 * (1.0 - t / tb)) - 1.0)
                      + 8.1328e-3 * (10.0 ** (-3.49149 * (tb / t - 1.0)) - 1.0) + np.log10(1013.246)) * 100.0
    assert np.max(np.abs(es / es_ref - 1.0)) < 1e-14
    # cp_exp: the optics' exp without OCML's range guards, over and far beyond the log-extinction range it serves
    xe = np.concatenate([rng.uniform(-40.0, 40.0, 400000), rng.uniform(-700.0, 700.0, 100000), np.array([0.0, -0.0, 1.0, -1.0, 1e-300])])
    ye = eng.math_probe("exp", eng.dev(xe)).cpu().numpy()
    re_ = np.exp(xe)
    assert (np.abs(ye - re_) / np.spacing(re_)).max() <= 2.0


@pytest.mark.parametrize("pver,ncol", [(32, 6075), (70, 777), (32, 48602), (128, 130), (4, 50)])
def test_zm_conv_tend_fused_tail_equals_separate_launches(pver, ncol, monkeypatch):
    """A small rank runs the tail of zm_conv_tend as three launches (geopotential_t || zm_conv_evap || momtran as heterogeneous
    blocks of one launch, the last two updates as one, the last two geopotential_t as one: ZmTailSmall) instead of nine; every
    output, state1 and history field included, bit for bit the same either way, at sizes either side of the switch"""
    eng = engine(cases.default_params(pver))
    st = synth.make_state(ncol, pver, 909, moist_boost=1.1)
    d = eng.dev
    S = {k: d(st[k]) for k in abi.ZM_TEND_IN if k != "ps"}
    S["ps"] = d(st["pint"][:, -1].copy())
    monkeypatch.setenv("COLPHYS_TAIL_FUSED", "0")
    a = to_np(eng.zm_conv_tend(S, synth.ZTODT, hist=True))
    monkeypatch.setenv("COLPHYS_TAIL_FUSED", "1")
    b = to_np(eng.zm_conv_tend(S, synth.ZTODT, hist=True))
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    if pver >= 26:
        assert a["lengath"].sum() > 0 and np.abs(a["ptend_u"]).max() > 0 and np.abs(a["evapcdp"]).max() > 0


@pytest.mark.parametrize("pver,ncol", [(32, 6075), (70, 777), (32, 30011)])
def test_uniform_rair_zvir_equal_the_fields(pver, ncol, monkeypatch):
    """rairv / zvirv hold one value each unless WACCM-X varies the composition (physics_types.F90:393-415): NULL for both fields
    makes the engine use its own constants -- two streams less per geopotential_t -- with the same bits as fields filled with
    them: geopotential_t of all three dycore branches, zm_conv_tend with the staged and with the fused tail; one of the two alone
    is an argument error"""
    eng = engine(cases.default_params(pver))
    st = synth.make_state(ncol, pver, 2024, moist_boost=1.1)
    d = eng.dev
    c = synth.PHYSCONST
    assert np.all(st["rair"] == c["rair"]) and np.all(st["zvir"] == c["zvir"])
    ga = [d(x) for x in cases.geo_args(st)[:6]]
    for dycore in (abi.DYCORE_EUL, abi.DYCORE_FV, abi.DYCORE_SE):
        a = eng.geopotential_t(*ga, d(st["t"]), d(st["q"]), d(st["rair"]), c["gravit"], d(st["zvir"]), dycore=dycore)
        b = eng.geopotential_t(*ga, d(st["t"]), d(st["q"]), None, c["gravit"], None, dycore=dycore)
        assert eng.torch.equal(a[0], b[0]) and eng.torch.equal(a[1], b[1]), dycore
    with pytest.raises(RuntimeError):
        eng.geopotential_t(*ga, d(st["t"]), d(st["q"]), d(st["rair"]), c["gravit"], None)
    S = {k: d(st[k]) for k in abi.ZM_TEND_IN if k != "ps"}
    S["ps"] = d(st["pint"][:, -1].copy())
    U = {k: v for k, v in S.items() if k not in ("rair", "zvir")}
    for fused in ("0", "1"):
        monkeypatch.setenv("COLPHYS_TAIL_FUSED", fused)
        a = to_np(eng.zm_conv_tend(S, synth.ZTODT, hist=True))
        b = to_np(eng.zm_conv_tend(U, synth.ZTODT, hist=True))
        for k in a:
            assert np.array_equal(a[k], b[k]), (fused, k)
    U["rair"] = S["rair"]
    with pytest.raises(RuntimeError):
        eng.zm_conv_tend(U, synth.ZTODT)


def test_lane_quads_pairs_and_single_lanes_give_the_same_bits(monkeypatch):
    """The dilute ascents with one, two or four lanes per column (colphys_common.h: lane pairs split the independent
    transcendentals of a function evaluation, lane quads also run the two opening evaluations of every inversion side by side)
    are the same operations on the same values: every output of zm_convr identical, at a strong-scaled share's size"""
    pver, ncol = 32, 6075
    eng = engine(cases.default_params(pver))
    st = synth.make_state(ncol, pver, 4404, moist_boost=1.1)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    outs = {}
    for lpc in ("1", "2", "4"):
        monkeypatch.setenv("COLPHYS_LPC_TRIG", lpc)
        monkeypatch.setenv("COLPHYS_LPC_ASC", lpc)
        outs[lpc] = run_zm(eng, args)
    assert outs["1"]["lengath"].sum() > 0.1 * ncol
    for lpc in ("2", "4"):
        for k, v in outs["1"].items():
            assert np.array_equal(outs[lpc][k], v), (lpc, k)


def test_trigger_block_shapes_give_the_same_bits(monkeypatch):
    """the trigger's blocks of one, two or four wave pairs (launch_zm_convr picks by column count: which waves share a SIMD)
    are a placement matter only; a mid-size share, where the choice changed in round 3, against the C oracle on a sample"""
    pver, ncol = 32, 24301
    eng = engine(cases.default_params(pver))
    st = synth.make_state(ncol, pver, 515, moist_boost=1.1)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    outs = {}
    for g in ("1", "2", "4"):
        monkeypatch.setenv("COLPHYS_TRIG_GROUPS", g)
        outs[g] = run_zm(eng, args)
    monkeypatch.delenv("COLPHYS_TRIG_GROUPS")
    outs["default"] = run_zm(eng, args)
    for g in ("1", "2", "default"):
        for k, v in outs["4"].items():
            assert np.array_equal(outs[g][k], v), (g, k)
    n = 1600                                    # whole chunks
    from c_oracle import COracle
    ref = COracle(cases.default_params(pver)).zm_convr(*[a[:n] if hasattr(a, "shape") and a.shape[:1] == (ncol,) else a for a in args])
    for k in ("heat", "qtnd", "prec", "cape"):
        assert_close("zm_convr." + k, outs["default"][k][:n], ref[k][:n])
    assert np.array_equal(outs["default"]["lengath"][:n // cases.PCOLS], ref["lengath"])
