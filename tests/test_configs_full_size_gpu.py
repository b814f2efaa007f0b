"""Every BASELINE.json configuration at its stated size on the GPU (configs 2-5; config 1 is the CPU plumbing case):

  config 3  geopotential_t + ZM, ne120-sized 777 602 columns x L32: every dycore branch of geopotential_t against the C
            restatement on ALL columns; the ZM chain at that size with a chunk-aligned sample against the C restatement
  config 5  48 602 columns x L70: the whole zm_conv_tend chain, momtran, modal_aero_sw / lw and the SW diagnostics at full
            size -- a chunk-aligned sample against the C restatement, the rest through size-independent properties
            (column permutation, linearity in the winds)
  configs 2 / 4  the substep bench.py times (cam_nor_physics_amd/substep.py: zm_conv_tend || optics on two streams, captured
            into a HIP graph, replayed) at 48 602 x L32 with the bench's own workload: a 2 000-column sample against the C
            restatement at 1e-10, graph replay == eager launches bit for bit, and a rank's share of the static split
            (config 4's per-GPU slab) equal to the same columns of the whole slab.

Tolerance 1e-10 relative per column profile (tests/util.py); integer and index outputs identical.
"""
import numpy as np
import pytest

import cases
from cam_nor_physics_amd import abi, synth
from cam_nor_physics_amd.dist import shard_range
from test_parity_gpu import _assert_swd, check_zm, engine, run_zm, to_np
from util import assert_close, gathered_mask

pytestmark = pytest.mark.gpu

NE120 = 777602
NE30 = 48602
_cache = {}


def ne120_state():
    if "ne120" not in _cache:
        _cache["ne120"] = synth.make_state(NE120, 32, 120120)
    return _cache["ne120"]


# ---------------------------------------------------------------------------------------------------
# config 3
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dycore,nspecies", [(abi.DYCORE_FV, 1), (abi.DYCORE_SE, 2), (abi.DYCORE_EUL, 1)])
def test_config3_geopotential_t_at_ne120_size_vs_c_oracle(dycore, nspecies):
    """777 602 columns (24 300 whole tiles of 32 and one of 2): zi, zm of every column against the C restatement
    (geopotential.F90:153-310); the hydrostatic scan is the same IEEE operations in the same order, so bit for bit"""
    from c_oracle import COracle
    pver = 32
    params = cases.default_params(pver)
    eng, co = engine(params), COracle(params)
    st = ne120_state()
    c = synth.PHYSCONST
    q = np.ascontiguousarray(np.stack([st["q"], 0.1 * st["q"]]))
    ga = cases.geo_args(st)
    zi_r, zm_r = co.geopotential_t(*ga, q, st["rair"], c["gravit"], st["zvir"], dycore=dycore, nspecies=nspecies)
    d = eng.dev
    zi, zm = eng.geopotential_t(*[d(x) for x in ga], d(q), d(st["rair"]), c["gravit"], d(st["zvir"]), dycore=dycore, nspecies=nspecies)
    zi, zm = zi.cpu().numpy(), zm.cpu().numpy()
    assert_close("geopotential_t.zi", zi, zi_r)
    assert_close("geopotential_t.zm", zm, zm_r)
    assert np.array_equal(zi, zi_r) and np.array_equal(zm, zm_r)
    assert (zi[:, :-1] > zi[:, 1:]).all() and not zi[:, -1].any()          # heights fall towards the surface, zi(pverp) = 0
    eng.torch.cuda.empty_cache()


def test_config3_zm_chain_at_ne120_size_sample_vs_c_oracle():
    """the kernel variants 777 602 columns select (one-wave trigger, two columns per plume wave): zm_conv_tend over all of
    them; the first 4 000 columns (250 whole chunks) against the C restatement's zm_convr, evaporation and momentum transport;
    every column finite, the gathered layout a faithful compaction"""
    from c_oracle import COracle
    pver, pcols, ns = 32, 16, 4000
    params = cases.default_params(pver)
    eng, co = engine(params), COracle(params)
    st = ne120_state()
    c = synth.PHYSCONST
    d = eng.dev
    S = {k: d(st[k]) for k in abi.ZM_TEND_IN if k != "ps"}
    S["ps"] = d(np.ascontiguousarray(st["pint"][:, -1]))
    o = eng.zm_conv_tend(S, synth.ZTODT, pcols=pcols)
    eng.torch.cuda.synchronize()
    assert not bool(o["status"].any())
    nd = int(o["lengath"].sum().item())
    assert 0.03 * NE120 < nd < 0.5 * NE120
    for k in ("ptend_s", "ptend_q", "ptend_u", "ptend_v", "prec", "snow", "cape", "zm1", "t1"):
        assert bool(eng.torch.isfinite(o[k]).all()), k
    head = {k: (v[:ns // pcols] if k == "lengath" else v[:ns]).cpu().numpy() for k, v in o.items()}
    del o, S
    eng.torch.cuda.empty_cache()
    s = {k: (v[:ns] if isinstance(v, np.ndarray) and v.shape[:1] == (NE120,) else v) for k, v in st.items()}
    _check_tend_sample(co, s, head, ns, pver, pcols)


def _check_tend_sample(co, s, o, ns, pver, pcols):
    """zm_conv_tend's outputs on a chunk-aligned sample against the C restatement composed the way zm_conv_tend composes the
    reference (zm_conv_intr.F90:662-836): zm_convr, update, geopotential_t, zm_conv_evap, update, momtran, update"""
    c, dt, qmin = synth.PHYSCONST, synth.ZTODT, 1.0e-12
    z = co.zm_convr(*cases.zm_args_from_state(s, s["zi"], s["zm"]))
    assert np.array_equal(o["lengath"], z["lengath"])
    m = gathered_mask(z["lengath"], ns, pcols)
    for k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg", "ideep"):
        assert not np.any(o[k][~m]), k
        assert_close("zm_conv_tend." + k, o[k][m], z[k][m])
    for k in ("ql", "rprd", "dlf", "cape", "rliq", "rice", "jctop", "jcbot"):
        assert_close("zm_conv_tend." + k, o[k], z[k])
    assert_close("zm_conv_tend.mconzm", o["mconzm"], z["mcon"] * 100.0 / c["gravit"])
    t1 = s["t"] + z["heat"] * dt / c["cpair"]
    q1 = np.maximum(s["q"] + z["qtnd"] * dt, qmin)
    e = co.zm_conv_evap(t1, s["pmid"], s["pdel"], q1, s["landfrac"], z["rprd"], s["cld"], dt, z["prec"])
    for k, r in (("prec", e["prec"]), ("snow", e["snow"]), ("evapcdp", e["tend_q"]), ("flxprec", e["flxprec"]), ("flxsnow", e["flxsnow"])):
        assert_close("zm_conv_tend." + k, o[k], r)
    t2 = t1 + e["tend_s"] * dt / c["cpair"]
    q2 = np.maximum(q1 + e["tend_q"] * dt, qmin)
    mo = co.momtran(s["u"], s["v"], z["mu"], z["md"], z["du"], z["eu"], z["ed"], z["dp"], z["dsubcld"], z["jt"], z["maxg"], z["ideep"],
                    z["lengath"], dt)
    assert_close("zm_conv_tend.ptend_s", o["ptend_s"], (z["heat"] + e["tend_s"]) + mo["seten"])
    assert_close("zm_conv_tend.ptend_q", o["ptend_q"], z["qtnd"] + e["tend_q"])
    assert_close("zm_conv_tend.ptend_u", o["ptend_u"], mo["dqdt"][0])
    assert_close("zm_conv_tend.ptend_v", o["ptend_v"], mo["dqdt"][1])
    t3 = t2 + mo["seten"] * dt / c["cpair"]
    assert_close("zm_conv_tend.t1", o["t1"], t3)
    assert_close("zm_conv_tend.q1", o["q1"], q2)
    assert_close("zm_conv_tend.u1", o["u1"], s["u"] + mo["dqdt"][0] * dt)
    zi3, zm3 = co.geopotential_t(*cases.geo_args(s)[:6], t3, q2, s["rair"], c["gravit"], s["zvir"])
    assert_close("zm_conv_tend.zi1", o["zi1"], zi3)
    assert_close("zm_conv_tend.zm1", o["zm1"], zm3)
    return z


# ---------------------------------------------------------------------------------------------------
# config 5
# ---------------------------------------------------------------------------------------------------
def test_config5_l70_zm_conv_tend_and_momtran_at_full_size():
    """48 602 x L70: the whole chain; 1 008 columns (63 chunks) against the C restatement; momtran at full size: doubling the
    winds doubles the wind tendencies, the pressure-gradient terms and the in-cloud winds bit for bit (momtran is linear in
    the winds, zm_conv.F90:2501-2627; a power of two commutes with every rounding) and quadruples the dissipative heating"""
    from c_oracle import COracle
    pver, pcols, ns = 70, 16, 1008
    params = cases.default_params(pver)
    eng, co = engine(params), COracle(params)
    st = synth.make_state(NE30, pver, 5070)
    d = eng.dev
    S = {k: d(st[k]) for k in abi.ZM_TEND_IN if k != "ps"}
    S["ps"] = d(np.ascontiguousarray(st["pint"][:, -1]))
    o = to_np(eng.zm_conv_tend(S, synth.ZTODT, pcols=pcols))
    assert not o["status"].any()
    nd = int(o["lengath"].sum())
    assert 0.03 * NE30 < nd < 0.5 * NE30
    head = {k: (v[:ns // pcols] if k == "lengath" else v[:ns]) for k, v in o.items()}
    s = {k: (v[:ns] if isinstance(v, np.ndarray) and v.shape[:1] == (NE30,) else v) for k, v in st.items()}
    _check_tend_sample(co, s, head, ns, pver, pcols)
    # momtran at full size on the gathered fluxes the chain produced
    g = [d(o[k]) for k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld")] + [d(o[k]) for k in ("jt", "maxg", "ideep", "lengath")]
    w = np.stack([st["u"], st["v"]])
    m1 = to_np(eng.momtran(d(w), *g, synth.ZTODT, pcols=pcols))
    m2 = to_np(eng.momtran(d(2.0 * w), *g, synth.ZTODT, pcols=pcols))
    for k in ("dqdt", "pguall", "pgdall", "icwu", "icwd"):
        assert np.isfinite(m1[k]).all() and np.array_equal(m2[k], 2.0 * m1[k]), k
    assert np.array_equal(m2["seten"], 4.0 * m1["seten"])
    assert np.array_equal(m1["dqdt"][0], o["ptend_u"]) and np.array_equal(m1["dqdt"][1], o["ptend_v"])
    deep_cols = np.flatnonzero(o["freqzm"] > 0)
    assert len(deep_cols) == nd and np.abs(m1["dqdt"][:, deep_cols]).max() > 0
    quiet = np.ones(NE30, bool); quiet[deep_cols] = False
    assert not m1["dqdt"][:, quiet].any() and not m1["seten"][quiet].any()


def test_config5_l70_optics_and_sw_diagnostics_at_full_size():
    """48 602 x L70: modal_aero_sw / lw and the SW history diagnostics; the first 1 000 columns against the C restatement,
    and a permutation of all the columns permutes every output bit for bit (cells and columns are independent)"""
    from c_oracle import COracle
    pver, ns = 70, 1000
    params = cases.default_params(pver)
    eng, co = engine(params), COracle(params)
    props = synth.make_aero_props(57)
    st = synth.make_state(NE30, pver, 57)
    aer = synth.make_aero_state(NE30, pver, props, 57)
    eng.aero_tables_set(props, synth.PHYSCONST)
    co.aero_tables_set(props, synth.PHYSCONST)
    d = eng.dev

    def run(idx):
        a = (d(st["pdeldry"][idx]), d(aer["specmmr"][:, :, idx]), d(aer["dgnumwet"][:, idx]), d(aer["qaerwat"][:, idx]))
        vis = eng.empty(3, eng.tables.nmodes, len(idx), pver)
        sw = eng.modal_aero_sw(*a, vis=vis)
        dg = eng.modal_aero_sw_diag(a[0], d(st["pmid"][idx]), d(st["t"][idx]), a[1], a[2], a[3], d(aer["troplevchem"][idx].astype(np.int32)),
                                    sw["tauxar"], vis=vis)
        lw = eng.modal_aero_lw(*a)
        r = {**{"sw." + k: v for k, v in sw.items()}, "lw.tauxar": lw["tauxar"], "lw.status": lw["status"], "swd.lev": dg["lev"], "swd.col": dg["col"]}
        return to_np(r)

    ident = np.arange(NE30)
    o = run(ident)
    assert not o["sw.status"].any() and not o["lw.status"].any()
    s = {k: st[k][:ns] for k in ("pdeldry", "pmid", "t")}
    a = {k: (aer[k][:ns] if k == "troplevchem" else aer[k][..., :ns, :]) for k in aer}
    sw_r, lw_r, dg_r = co.modal_aero_sw(s, a), co.modal_aero_lw(s, a), co.modal_aero_sw_diag(s, a)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, o["sw." + k][:ns], sw_r[k])
    assert_close("modal_aero_lw.tauxar", o["lw.tauxar"][:ns], lw_r["tauxar"])
    _assert_swd(dict(lev=o["swd.lev"][:, :ns], col=o["swd.col"][:, :ns]), dg_r["lev"], dg_r["col"], eng.tables.nmodes)
    perm = np.random.default_rng(70).permutation(NE30)
    p = run(perm)
    for k in ("sw.tauxar", "sw.wa", "sw.ga", "sw.fa", "lw.tauxar"):
        assert np.array_equal(p[k], o[k][perm]), k
    assert np.array_equal(p["swd.lev"], o["swd.lev"][:, perm]) and np.array_equal(p["swd.col"], o["swd.col"][:, perm])
    assert (o["sw.tauxar"][:, :, 1:] > 0).all() and (o["lw.tauxar"] >= 0).all()
    eng.torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------
# configs 2 and 4: the timed path of bench.py itself
# ---------------------------------------------------------------------------------------------------
def _bench_workload(pver, moist_boost=1.1):
    props = synth.make_aero_props(synth.DEFAULT_SEED)
    st = synth.make_state(NE30, pver, synth.DEFAULT_SEED, moist_boost=moist_boost)
    aer = synth.make_aero_state(NE30, pver, props, synth.DEFAULT_SEED)
    return st, aer, props


def test_bench_substep_graph_replay_vs_c_oracle_and_eager():
    """the object bench.py times, built the way bench.py builds it (two streams, one HIP graph, the bench's seed and
    moist_boost), at 48 602 x L32: after graph replays into NaN-filled outputs
      * a 2 000-column sample (125 chunks) of every output equals the C restatement to 1e-10 -- ZM chain, optics, SW diagnostics;
      * the replay's outputs equal the eager launches' bit for bit (every output array, not a checksum);
      * the convecting fraction is inside SURVEY.md 8d's 10-30 % band"""
    from c_oracle import COracle
    from cam_nor_physics_amd.substep import ResidentSubstep
    pver, pcols, ns = 32, 16, 2000
    params = cases.default_params(pver)
    eng, co = engine(params), COracle(params)
    st, aer, props = _bench_workload(pver)
    ss = ResidentSubstep(eng, st, aer, props, pcols=pcols).capture()
    ss.clear_outputs()
    ss.step(); ss.step()
    eng.torch.cuda.synchronize()
    got = to_np(ss.outputs())
    frac = got["zm.lengath"].sum() / NE30
    assert 0.10 <= frac <= 0.30, frac
    assert not got["zm.status"].any() and not got["sw.status"].any() and not got["lw.status"].any()
    # eager launches into cleared buffers: identical bits
    ss.clear_outputs()
    ss.launch()
    eng.torch.cuda.synchronize()
    eager = to_np(ss.outputs())
    for k in got:
        assert not (got[k].dtype == np.float64 and np.isnan(got[k]).any()), f"{k}: elements no kernel wrote"
        assert np.array_equal(got[k], eager[k]), k
    ss.release_graph()
    # the sample against the C restatement
    s = {k: (v[:ns] if isinstance(v, np.ndarray) and v.shape[:1] == (NE30,) else v) for k, v in st.items()}
    zm = {k[3:]: (v[:ns // pcols] if k == "zm.lengath" else v[:ns]) for k, v in got.items() if k.startswith("zm.")}
    z = _check_tend_sample(co, s, zm, ns, pver, pcols)
    assert z["lengath"].sum() > 0.1 * ns
    co.aero_tables_set(props, synth.PHYSCONST)
    a = {k: (aer[k][:ns] if k == "troplevchem" else aer[k][..., :ns, :]) for k in aer}
    sw_r, lw_r, dg_r = co.modal_aero_sw(s, a), co.modal_aero_lw(s, a), co.modal_aero_sw_diag(s, a)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, got["sw." + k][:ns], sw_r[k])
    assert_close("modal_aero_lw.tauxar", got["lw.tauxar"][:ns], lw_r["tauxar"])
    _assert_swd(dict(lev=got["swd.lev"][:, :ns], col=got["swd.col"][:, :ns]), dg_r["lev"], dg_r["col"], eng.tables.nmodes)
    _cache["bench_l32"] = got


def test_config4_static_split_share_equals_the_whole_slab():
    """config 4 on the one GPU a test has: rank 3's share of the 8-way static split of the 48 602-column slab (chunk-aligned,
    dist.shard_range), run as bench.py --gpus 8 runs it, equals the same columns of the one-GPU run bit for bit -- per-column
    outputs, the gathered sets of its chunks, optics and diagnostics -- so the union of the eight shares is the whole slab"""
    from cam_nor_physics_amd.substep import ResidentSubstep
    pver, pcols, world, rank = 32, 16, 8, 3
    if "bench_l32" not in _cache:
        test_bench_substep_graph_replay_vs_c_oracle_and_eager()
    whole = _cache["bench_l32"]
    eng = engine(cases.default_params(pver))
    st, aer, props = _bench_workload(pver)
    covered = 0
    for r in range(world):
        lo, hi = shard_range(NE30, world, r, pcols)
        assert lo % pcols == 0 and lo == covered
        covered = hi
    assert covered == NE30
    lo, hi = shard_range(NE30, world, rank, pcols)
    s = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (NE30,) else v) for k, v in st.items()}
    s["ncol"] = hi - lo
    a = {k: (aer[k][lo:hi] if k == "troplevchem" else aer[k][..., lo:hi, :]) for k in aer}
    ss = ResidentSubstep(eng, s, a, props, pcols=pcols).capture()
    ss.clear_outputs()
    ss.step()
    eng.torch.cuda.synchronize()
    part = to_np(ss.outputs())
    ss.release_graph()
    for k, v in part.items():
        w = whole[k]
        if k == "zm.lengath":
            ref = w[lo // pcols:(hi + pcols - 1) // pcols]
        elif k in ("swd.lev", "swd.col"):
            ref = w[:, lo:hi]
        else:
            ref = w[lo:hi]
        assert np.array_equal(v, ref), k
