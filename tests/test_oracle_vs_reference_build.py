"""CPU test: the C restatement (oracle/colphys_oracle.c) against the reference's own Fortran built in oracle/_ref, over
namelist / sounding regimes the committed golden fixtures do not visit (PBL-mixed parcels, no_deep_pbl, 1..5 CIN regions,
tiedke_add, capelmt, tau, dmpdz, momentum / autoconversion / evaporation coefficients; cold, hot, dry and very moist
soundings).  This widens what pins the oracle beyond tests/golden; it runs wherever oracle/_ref has been built
(`__graft_entry__.build()` builds it when /root/reference is present) and is skipped elsewhere."""
import numpy as np
import pytest

import cases
from c_oracle import COracle
from cam_nor_physics_amd import abi, synth
from ref_oracle import RefOracle, have_ref
from util import assert_close, gathered_mask

REGIMES = 12


def _regime(r):
    rng = np.random.default_rng(1000 + r)
    pver = (32, 70)[r % 2]
    ncol = int(rng.integers(150, 500))
    nl = dict(synth.ZM_NAMELIST)
    nl.update(parcel_pbl=int(rng.integers(0, 2)), no_deep_pbl=int(rng.integers(0, 2)), num_cin=int(rng.integers(1, 6)),
              tiedke_add=float(rng.choice([0.0, 0.5, 1.0])), capelmt=float(rng.choice([30.0, 70.0, 150.0])),
              tau=float(rng.choice([1800.0, 3600.0, 7200.0])), dmpdz=float(rng.choice([-0.5e-3, -1.0e-3, -2.0e-3])),
              momcu=float(rng.uniform(0.2, 0.9)), momcd=float(rng.uniform(0.2, 0.9)),
              c0_lnd=float(rng.choice([0.0035, 0.0075])), c0_ocn=float(rng.choice([0.0035, 0.03, 0.045])),
              ke=float(rng.choice([1.0e-6, 5.0e-6])))
    ts = [(255.0, 285.0), (275.0, 303.0), (295.0, 310.0)][int(rng.integers(0, 3))]
    rh = [(0.1, 0.6), (0.3, 0.95), (0.8, 0.99)][int(rng.integers(0, 3))]
    return pver, ncol, nl, ts, rh, int(rng.integers(1, 1 << 30))


@pytest.mark.parametrize("r", range(REGIMES))
def test_c_restatement_vs_reference_build_over_regimes(r):
    pver, ncol, nl, ts, rh, seed = _regime(r)
    if not have_ref(cases.PCOLS, pver):
        pytest.skip("oracle/_ref is not built here")
    limcnv = synth.limcnv_from_pref(pver)
    ref = RefOracle(cases.PCOLS, pver)
    ref.zm_convi(limcnv, nl)
    co = COracle(abi.make_zm_params(synth.PHYSCONST, nl, limcnv))
    st = synth.make_state(ncol, pver, seed, ts_range=ts, rh_range=rh)
    c = synth.PHYSCONST
    ga = cases.geo_args(st)
    zi, zm = co.geopotential_t(*ga, st["q"], st["rair"], c["gravit"], st["zvir"], dycore=abi.DYCORE_FV)
    zi_r, zm_r = ref.geopotential_t(*ga, st["q"], st["rair"], c["gravit"], st["zvir"], dycore=abi.DYCORE_FV)
    assert np.array_equal(zi, zi_r) and np.array_equal(zm, zm_r)
    args = cases.zm_args_from_state(st, zi, zm)
    o, g = co.zm_convr(*args), ref.zm_convr(*args)
    assert np.array_equal(o["lengath"], g["lengath"])
    m = gathered_mask(g["lengath"], ncol, cases.PCOLS)
    for k in g:
        if k == "lengath":
            continue
        a, b = o[k], g[k]
        if k in abi.ZM_CONVR_GATHERED or k == "ideep":
            a, b = a[m], b[m]
        assert_close("zm_convr." + k, a, b)
    dt = synth.ZTODT
    t1 = st["t"] + g["heat"] / c["cpair"] * dt
    q1 = st["q"] + g["qtnd"] * dt
    ea = (t1, st["pmid"], st["pdel"], q1, st["landfrac"], g["rprd"], st["cld"], dt, g["prec"])
    e, e_r = co.zm_conv_evap(*ea), ref.zm_conv_evap(*ea)
    for k, v in e_r.items():
        assert_close("zm_conv_evap." + k, e[k], v)
    ma = (st["u"], st["v"], g["mu"], g["md"], g["du"], g["eu"], g["ed"], g["dp"], g["dsubcld"], g["jt"], g["maxg"], g["ideep"],
          g["lengath"], dt)
    mo, mo_r = co.momtran(*ma), ref.momtran(*ma)
    for k, v in mo_r.items():
        assert_close("momtran." + k, mo[k], v)


def test_the_regimes_visit_what_they_claim():
    regs = [_regime(r) for r in range(REGIMES)]
    assert {x[2]["parcel_pbl"] for x in regs} == {0, 1} and {x[2]["no_deep_pbl"] for x in regs} == {0, 1}
    assert len({x[2]["num_cin"] for x in regs}) >= 4 and len({x[3] for x in regs}) == 3 and len({x[4] for x in regs}) == 3


@pytest.mark.parametrize("pver,nspec_m,top_lev,edges", [
    (32, (5,), 1, False), (70, (1,), 1, False), (32, (2, 6), 3, False), (70, (8, 1, 3), 1, True), (32, (1, 1, 1, 1), 1, False),
    (32, (3, 2, 4, 1, 5), 2, True), (70, (2,) * 9, 1, False), (32, synth.NSPEC_M, 1, True)])
def test_c_restatement_optics_vs_reference_build(pver, nspec_m, top_lev, edges):
    """modal_aero_sw / modal_aero_lw / the SW history fields for mode and species counts and side branches the fixtures do
    not hold: one to nine modes, one to eight species, clim_modal_aero_top_lev > 1, and (edges) negative aerosol water,
    particles beyond the size table on both sides, water-dominated and almost empty cells, the tropopause above the top and
    below the surface.  No input makes an optical depth negative: the reference would end the process (endrun)."""
    if not have_ref(cases.PCOLS, pver):
        pytest.skip("oracle/_ref is not built here")
    ncol = 120
    ref = RefOracle(cases.PCOLS, pver)
    co = COracle(cases.default_params(pver))
    props = synth.make_aero_props(40 + len(nspec_m), nspec_m=nspec_m)
    props["top_lev"] = top_lev
    st = synth.make_state(ncol, pver, 41)
    aer = synth.make_aero_state(ncol, pver, props, 41)
    aer["troplevchem"] = aer["troplevchem"].astype(np.int32)
    if edges:
        rng = np.random.default_rng(8)
        qw, dg, mm = aer["qaerwat"], aer["dgnumwet"], aer["specmmr"]
        qw[:, ::7, :] *= -1.0
        qw[:, 3::11, :] *= 1.0e3
        dg[:, 1::5, :] = 10.0 ** rng.uniform(-4.6, -3.5, dg[:, 1::5, :].shape)
        dg[:, 2::9, :] = 10.0 ** rng.uniform(-9.5, -8.2, dg[:, 2::9, :].shape)
        mm[:, :, 4::13, :] *= 1.0e-8
        aer["troplevchem"][::5] = 0
        aer["troplevchem"][1::7] = pver + 3
    ref.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    sw, sw_r = co.modal_aero_sw(st, aer), ref.modal_aero_sw(st, aer)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, sw[k], sw_r[k])
    assert_close("modal_aero_lw.tauxar", co.modal_aero_lw(st, aer)["tauxar"], ref.modal_aero_lw(st, aer)["tauxar"])
    dg_c, dg_r = co.modal_aero_sw_diag(st, aer), ref.modal_aero_sw_diag(st, aer)
    lev_names, col_names = cases.swd_field_names(len(nspec_m))
    got_col = cases.swd_reference_ssavis(dg_c["col"])          # SSAVISdn: the restatement returns the sum, the reference the quotient
    for i, nm in enumerate(lev_names):
        assert_close(nm, dg_c["lev"][i], dg_r["lev"][i])
    for i, nm in enumerate(col_names):
        assert_close(nm, got_col[i], dg_r["col"][i])
