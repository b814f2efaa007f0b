"""Drop-in test of the ISO_C_BINDING shim (cam-nor-physics_amd/fortran/): the Fortran modules `zm_conv` and
`geopotential` built from the shim are driven by the SAME caller source that drives the reference's modules in
oracle/_ref (oracle/ref_wrap_zm.F90: chunk loop, (pcols,pver) arrays with +inf padding, zm_convi / zm_convr /
zm_conv_evap / momtran / geopotential_t calls with the reference's argument lists) and must reproduce the
reference's golden outputs."""
import os

import numpy as np
import pytest

import cases
from cam_nor_physics_amd import abi, synth
from util import assert_close, gathered_mask

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shim(pver):
    from ref_oracle import RefOracle
    path = os.path.join(ROOT, "cam-nor-physics_amd", "_build", f"libcolphys_shim_p16_l{pver}.so")
    assert os.path.exists(path), "build the shim: make -C cam-nor-physics_amd/fortran"
    return RefOracle(16, pver, lib_path=path)


@pytest.mark.parametrize("name", ["zm_l32", "zm_l70", "zm_l32_pbl"])
def test_fortran_caller_on_hip_engine_matches_reference_golden(name):
    g = cases.load_golden(name)
    n, pver = g["in_t"].shape
    f = shim(pver)
    f.zm_convi(int(g["limcnv"]), cases.golden_namelist(g))          # -> colphys_init through the shim's zm_convi
    q2 = np.stack([g["in_q"], 0.1 * g["in_q"]])
    for dy, tag in ((0, "eul"), (1, "fv"), (2, "se")):
        zi, zm = f.geopotential_t(g["in_lnpint"], g["in_lnpmid"], g["in_pint"], g["in_pmid"], g["in_pdel"], g["in_rpdel"],
                                  g["in_t"], q2, g["in_rair"], synth.PHYSCONST["gravit"], g["in_zvir"], dycore=dy,
                                  nspecies=2 if dy == 2 else 1)
        assert_close(f"geopotential_t[{tag}].zi", zi, g[f"geo_{tag}_zi"])
        assert_close(f"geopotential_t[{tag}].zm", zm, g[f"geo_{tag}_zm"])
    o = f.zm_convr(*cases.zm_args_from_golden(g))
    assert np.array_equal(o["lengath"], g["zm_lengath"])
    m = gathered_mask(g["zm_lengath"], n, 16)
    for k in o:
        if k == "lengath":
            continue
        a, b = o[k], g["zm_" + k]
        if k in abi.ZM_CONVR_GATHERED or k == "ideep":
            a, b = a[m], b[m]
        assert_close("zm_convr." + k, a, b)
    e = f.zm_conv_evap(g["in_evap_t"], g["in_pmid"], g["in_pdel"], g["in_evap_q"], g["in_landfrac"], g["zm_rprd"], g["in_cld"],
                       synth.ZTODT, g["zm_prec"])
    for k, v in e.items():
        assert_close("zm_conv_evap." + k, v, g["evap_" + k])
    mo = f.momtran(g["in_u"], g["in_v"], g["zm_mu"], g["zm_md"], g["zm_du"], g["zm_eu"], g["zm_ed"], g["zm_dp"], g["zm_dsubcld"],
                   g["zm_jt"], g["zm_maxg"], g["zm_ideep"], g["zm_lengath"], synth.ZTODT)
    for k, v in mo.items():
        assert_close("momtran." + k, v, g["mom_" + k])
    cq, cf, cdp = cases.convtran_inputs(g["in_pdel"], g["in_q"], g["zm_ideep"], g["zm_lengath"], int(g["ct_seed"]))
    ct = f.convtran(cq, cf, g["zm_mu"], g["zm_md"], g["zm_du"], g["zm_eu"], g["zm_ed"], g["zm_dp"], cdp, g["zm_jt"], g["zm_maxg"],
                    g["zm_ideep"], g["zm_lengath"], cases.DOCONVTRAN, cases.ISDRY, synth.ZTODT)
    assert_close("convtran.dqdt", ct, g["ct_dqdt"])


@pytest.mark.parametrize("pver", [32, 70])
def test_fortran_geopotential_dse_on_hip_engine_matches_reference_golden(pver):
    """module geopotential is replaced whole: geopotential_dse (geopotential.F90:29) through the shim, bit for bit"""
    g = cases.load_golden(f"geo_dse_l{pver}")
    f = shim(pver)
    f.zm_convi(synth.limcnv_from_pref(pver), synth.ZM_NAMELIST)     # colphys_init
    for dy, tag in ((0, "eul"), (1, "fv"), (2, "se")):
        args, nsp, _ = cases.geo_dse_args_from_golden(g, tag, int(g["ncol"]), pver, int(g["seed"]), dy)
        t, zi, zm = f.geopotential_dse(*args, dycore=dy, nspecies=nsp)
        assert np.array_equal(t, g[f"{tag}_t"]) and np.array_equal(zi, g[f"{tag}_zi"]) and np.array_equal(zm, g[f"{tag}_zm"]), tag


@pytest.mark.parametrize("name", cases.AERO_GOLDEN)
def test_fortran_optics_caller_on_hip_engine_matches_reference_golden(name):
    """`module modal_aer_opt` built from the shim, driven by the caller source that drives the reference's module
    (oracle/ref_wrap_aero.F90: tables served through the rad_constituents / pbuf / pio accessors, chunk loop,
    modal_aer_opt_init / modal_aero_sw / modal_aero_lw with the reference's argument lists)."""
    g = cases.load_golden(name)
    n, pver = g["in_t"].shape
    f = shim(pver)
    f.aero_tables_set(cases.aero_props_from_golden(g), synth.PHYSCONST)   # -> modal_aer_opt_init of the shim
    st = {k: g["in_" + k] for k in ("pdeldry", "pmid", "t")}
    aer = {k: g["in_" + k] for k in ("specmmr", "dgnumwet", "qaerwat", "troplevchem")}
    sw = f.modal_aero_sw(st, aer)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("modal_aero_sw." + k, sw[k], g["sw_" + k])
    lw = f.modal_aero_lw(st, aer)
    assert_close("modal_aero_lw.tauxar", lw["tauxar"], g["lw_tauxar"])
    # the history fields the shim's modal_aero_sw hands to outfld, captured by the same cam_history stand-in that captured
    # the reference's: "dn" set and day-only twins, night columns filled, SSAVISdn with the reference's night quirk
    dg = f.modal_aero_sw_diag(st, aer, g["in_nite"])
    lev_names, col_names = cases.swd_field_names(len(cases.aero_props_from_golden(g)["nspec"]))
    for tag in ("", "_f"):
        for i, nm in enumerate(lev_names):
            assert_close(nm + tag, dg["lev" + tag][i], g["swd_lev" + tag][i])
        for i, nm in enumerate(col_names):
            assert_close(nm + tag, dg["col" + tag][i], g["swd_col" + tag][i])


@pytest.mark.parametrize("name", cases.AERO_GOLDEN)
def test_fortran_optics_whole_rank_dispatch_matches_reference_golden(name):
    """The optics' half of the host face from Fortran (modal_aero_rank_* of the shim's `module modal_aer_opt`,
    tests/fortran/hostface_aero_driver.F90): the chunks' fields fetched through the same accessors as the per-chunk calls and
    pushed into the records, ONE dispatch for all chunks, results and history fields pulled per chunk -- against the goldens of
    the reference's per-chunk modal_aero_sw / modal_aero_lw."""
    g = cases.load_golden(name)
    n, pver = g["in_t"].shape
    f = shim(pver)
    props = cases.aero_props_from_golden(g)
    f.aero_tables_set(props, synth.PHYSCONST)
    st = {k: g["in_" + k] for k in ("pdeldry", "pmid", "t")}
    aer = {k: g["in_" + k] for k in ("specmmr", "dgnumwet", "qaerwat", "troplevchem")}
    o = f.modal_aero_rank(st, aer, g["in_nite"])
    for k in ("tauxar", "wa", "ga", "fa"):
        assert_close("rank modal_aero_sw." + k, o[k], g["sw_" + k])
    assert_close("rank modal_aero_lw.tauxar", o["lw_tauxar"], g["lw_tauxar"])
    lev_names, col_names = cases.swd_field_names(len(props["nspec"]))
    for tag in ("", "_f"):
        for i, nm in enumerate(lev_names):
            assert_close("rank " + nm + tag, o["lev" + tag][i], g["swd_lev" + tag][i])
        for i, nm in enumerate(col_names):
            assert_close("rank " + nm + tag, o["col" + tag][i], g["swd_col" + tag][i])
    # and bit for bit what the per-chunk calls of the same shim return (same kernels, other route)
    sw = f.modal_aero_sw(st, aer)
    for k in ("tauxar", "wa", "ga", "fa"):
        assert np.array_equal(o[k], sw[k]), k
    assert np.array_equal(o["lw_tauxar"], f.modal_aero_lw(st, aer)["tauxar"])


_HALT_SCRIPT = r"""
import os, sys
import numpy as np
root = sys.argv[1]
for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
    sys.path.insert(0, p)
import cases
from cam_nor_physics_amd import synth
from ref_oracle import RefOracle
case, pver, ncol = sys.argv[2], 32, 40
f = RefOracle(16, pver, lib_path=os.path.join(root, "cam-nor-physics_amd", "_build", "libcolphys_shim_p16_l32.so"))
f.zm_convi(synth.limcnv_from_pref(pver), synth.ZM_NAMELIST)
st = synth.make_state(ncol, pver, 5)
if case == "nan-sounding":
    a = list(cases.zm_args_from_state(st, st["zi"], st["zm"]))
    a[0] = a[0].copy(); a[0][21, :] = np.nan
    f.zm_convr(*a)
else:
    props = synth.make_aero_props(9)
    aer = synth.make_aero_state(ncol, pver, props, 9)
    f.aero_tables_set(props, synth.PHYSCONST)
    s = {k: st[k] for k in ("pdeldry", "pmid", "t")}
    if case in ("heavy-sw", "heavy-lw"):
        aer["specmmr"][:, :, 3, 10:14] *= 3.0e6
    if case == "negative-sw":
        s["pdeldry"] = s["pdeldry"].copy(); s["pdeldry"][17, 5] *= -1.0e4
    if case.endswith("lw"):
        for _ in range(int(sys.argv[3])):
            f.modal_aero_lw(s, aer)
    else:
        f.modal_aero_sw(s, aer)
print("RETURNED")
"""


@pytest.mark.parametrize("case,extra,halts,message", [
    ("nan-sounding", "", True, "**** ZM_CONV IENTHALPY: Tmix did not converge ****"),          # zm_conv.F90:5565
    ("heavy-sw", "", False, "WARNING: Aerosol optical depth is unreasonably high in this layer."),  # modal_aer_opt.F90:1002: no halt
    ("negative-sw", "", True, "*** halting in modal_aero_sw after nerr_dopaer ="),             # :1024-1026
    ("heavy-lw", "2", False, "WARNING: Aerosol optical depth is unreasonably high in this layer."),
    ("heavy-lw", "6", True, "*** halting in modal_aero_lw after nerr_dopaer ="),               # :1432: the count reached 1000
])
def test_shim_halts_where_the_reference_halts(case, extra, halts, message):
    """The conditions the reference ends the run on, through the shim's modules (endrun here is the stand-in that prints
    'ENDRUN: <message>' and stops the process, hence the child process): a failed inversion halts with the reference's text;
    an unreasonably LARGE optical depth is a warning that modal_aero_sw survives and modal_aero_lw survives until its
    counter, kept over the calls, reaches nerrmax_dopaer = 1000 (one heavy column of these inputs adds about 220 per call);
    a NEGATIVE one halts at once."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _HALT_SCRIPT, ROOT, case, extra], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert message in out, out[-2000:]
    if halts:
        assert r.returncode != 0 and "RETURNED" not in r.stdout and "ENDRUN:" in out, out[-2000:]
    else:
        assert r.returncode == 0 and "RETURNED" in r.stdout and "ENDRUN:" not in out, out[-2000:]


# ---------------------------------------------------------------------------------------------------
# the reference's OPERATOR files on the HIP engine, not a line changed
# ---------------------------------------------------------------------------------------------------
_shimintr = {}


def shimintr(pver, tab):
    """libcolphys_shimintr_*: /root/reference/physics/{physics_types,zm_conv_intr,convect_diagnostics}.F90 compiled unmodified
    against the shim's `module zm_conv` / `module geopotential` (cam-nor-physics_amd/fortran/Makefile, target intr)"""
    from ref_intr_oracle import RefIntrOracle
    path = os.path.join(ROOT, "cam-nor-physics_amd", "_build", f"libcolphys_shimintr_p16_l{pver}.so")
    assert os.path.exists(path), "build it where /root/reference exists: make -C cam-nor-physics_amd/fortran intr"
    if pver not in _shimintr:           # zm_conv_init allocates module state of zm_conv_intr once per process
        pe = synth.PTOP + synth.sigma_edges(pver) * (1.0e5 - synth.PTOP)
        _shimintr[pver] = RefIntrOracle(synth.ZM_NAMELIST, pe, 16, pver, lib_path=path, **tab)
    return _shimintr[pver]


@pytest.mark.parametrize("name", cases.INTR_GOLDEN)
def test_reference_zm_conv_tend_unmodified_on_hip_engine_matches_reference_golden(name):
    """zm_conv_intr.F90's zm_conv_tend and zm_conv_tend_2 -- the reference's own source: pbuf lookups, physics_state_copy,
    zm_convr, outfld, physics_ptend_sum, physics_update (whose geopotential_t is the shim's too), zm_conv_evap, momtran, convtran
    -- linked against the shim instead of zm_conv.F90 / geopotential.F90, against the fixtures the all-reference build wrote:
    ptend_all, every pbuf field, every outfld value, zm_conv_tend_2's tracer tendencies.  1e-10 relative."""
    g = cases.load_golden(name)
    pver = int(g["pver"])
    tab = {k[4:]: g[k] for k in ("tab_isdry", "tab_lq1", "tab_lq2", "tab_qmin")}
    f = shimintr(pver, tab)
    st = cases.intr_state_from_golden(g)
    o = f.zm_conv_tend(st, synth.ZTODT, q=g["in_q6"], fracis=g["in_fracis6"])
    r = {k: v for k, v in o.items() if k not in ("q_in", "fracis_in", "ptend_lq")}
    r["ptend_q"] = o["ptend_q"][:6]
    assert np.array_equal(o["ptend_lq"], g["tend_ptend_lq"])
    cases.assert_tend_matches_golden(r, g)
    assert set(r["hist"]) == {k[5:] for k in g if k.startswith("hist_")}
    pq, lq = f.zm_conv_tend_2(g["in_q6"], g["in_pdeldry"], g["in_fracis6"], *[o[k] for k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld",
                                                                                                   "jt", "maxg", "ideep")], synth.ZTODT)
    assert np.array_equal(lq, g["tend2_lq"])
    for m in range(6):
        assert_close(f"zm_conv_tend_2.ptend_q[{m}]", pq[m], g["tend2_ptend_q"][m])
    # physics_update of the reference's physics_types.F90 with the shim's geopotential_t inside it (physics_types.F90:436)
    pu = f.physics_update(st, synth.ZTODT, s=g["pu_tend_s"], q=g["pu_tend_q"], u=g["pu_tend_u"], v=g["pu_tend_v"])
    for k in ("t", "q", "u", "v", "zi", "zm", "s"):
        assert_close("physics_update." + k, pu[k], g["pu_" + k])
    # BASELINE config 4's oslo_aero_microp: the engine's drop-in module behind the same driver as the reference's empty one
    assert f.oslo_aero_microp() == dict(fields_flagged=0, components_allocated=0, use_oslo_aero=False, nbmodes=0)


@pytest.mark.parametrize("name", cases.INTR_GOLDEN)
def test_fortran_physics_state_face_whole_rank_dispatch_matches_reference_operator_golden(name):
    """The restructured chunk loop (fortran/colphys_hostface_mod.F90 driven by tests/fortran/hostface_driver.F90): one
    physics_state per chunk, built by the reference's own physics_state_alloc (+inf padding), pushed chunk by chunk, ONE
    dispatch for all of them, pulled into physics_ptend / zm_conv_tend's arguments / pbuf arrays -- against what the
    reference's zm_conv_tend wrote for the same columns (tests/golden intr_*): ptend_all, the surface and pbuf fields, the
    gathered sets.  The fixture has a chunk with no convecting column, chunks with several, and a ragged last chunk."""
    import ctypes as C
    g = cases.load_golden(name)
    pver, n = int(g["pver"]), int(g["ncol"])
    tab = {k[4:]: g[k] for k in ("tab_isdry", "tab_lq1", "tab_lq2", "tab_qmin")}
    f = shimintr(pver, tab)                              # zm_conv_init -> the shim's zm_convi -> colphys_init
    lib = f.lib
    st = cases.intr_state_from_golden(g)
    D = C.POINTER(C.c_double)
    I = C.POINTER(C.c_int)
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    nch = (n + 15) // 16
    o = {k: np.zeros((n, pver)) for k in ("ptend_s", "ptend_u", "ptend_v", "ptend_q", "ql", "rprd", "evapcdp", "dlf")}
    o.update({k: np.zeros((nch * 16, pver)) for k in ("mu", "md", "du", "eu", "ed", "dp")})
    o.update({k: np.zeros((n, pver + 1)) for k in ("mcon", "mconzm", "flxprec", "flxsnow")})
    o.update({k: np.zeros(n) for k in ("rliq", "rice", "jctop", "jcbot", "prec", "snow", "cape", "freqzm", "pcont", "pconb")})
    o["dsubcld"] = np.zeros(nch * 16)
    o.update({k: np.zeros(nch * 16, np.int32) for k in ("jt", "maxg", "ideep")})
    o["lengath"] = np.zeros(nch, np.int32)
    times = np.zeros(2)
    ins = [c(st[k]) for k in ("t", "u", "v", "q", "pmid", "pdel", "rpdel", "lnpmid", "zm", "pint", "lnpint", "zi", "phis")] + \
          [c(st["pint"][:, -1])] + [c(st[k]) for k in ("pblh", "tpert", "landfrac", "cld")]
    dp_ = lambda a: a.ctypes.data_as(D)
    lib.hfd_zm_conv_tend(C.c_int(n), C.c_double(synth.ZTODT), *[dp_(a) for a in ins],
                         *[dp_(o[k]) for k in ("ptend_s", "ptend_u", "ptend_v", "ptend_q", "mcon", "rliq", "rice", "jctop", "jcbot", "prec", "snow",
                                               "ql", "rprd", "evapcdp", "dlf", "mconzm", "flxprec", "flxsnow", "mu", "md", "du", "eu", "ed", "dp",
                                               "dsubcld")],
                         *[o[k].ctypes.data_as(I) for k in ("jt", "maxg", "ideep", "lengath")],
                         *[dp_(o[k]) for k in ("cape", "freqzm", "pcont", "pconb")], dp_(times))
    r = {k: (v[:n] if k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg", "ideep") else v) for k, v in o.items()}
    r["ptend_q"] = o["ptend_q"][None]                   # water vapour; cloud liquid / ice are convtran1's (zm_conv_tend_2_dev)
    r["hist"] = dict(CAPE=o["cape"], FREQZM=o["freqzm"], PCONVT=o["pcont"], PCONVB=o["pconb"], CMFMC_DP=o["mconzm"])
    cases.assert_tend_matches_golden(r, g)
    assert times[0] > 0 and times[1] < times[0]         # the dispatching thread mostly sleeps


_HF_HALT = r"""
import ctypes as C, os, sys
import numpy as np
root = sys.argv[1]
for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
    sys.path.insert(0, p)
import torch                                   # first: the process's HIP runtime is the one torch brings
import cases
from cam_nor_physics_amd import synth
from ref_intr_oracle import RefIntrOracle
pver, n = 32, 16 * 6 - 3
pe = synth.PTOP + synth.sigma_edges(pver) * (1.0e5 - synth.PTOP)
f = RefIntrOracle(synth.ZM_NAMELIST, pe, 16, pver, lib_path=os.path.join(root, "cam-nor-physics_amd", "_build", "libcolphys_shimintr_p16_l32.so"))
st = synth.make_state(n, pver, 5)
if sys.argv[2] == "nan":
    st["t"] = st["t"].copy(); st["t"][37, :] = np.nan
D, I = C.POINTER(C.c_double), C.POINTER(C.c_int)
c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
nch = (n + 15) // 16
mid, itf, col = (lambda: np.zeros((nch * 16, pver))), (lambda: np.zeros((n, pver + 1))), (lambda: np.zeros(nch * 16))
ins = [c(st[k]) for k in ("t", "u", "v", "q", "pmid", "pdel", "rpdel", "lnpmid", "zm", "pint", "lnpint", "zi", "phis")] + [c(st["pint"][:, -1])] + \
      [c(st[k]) for k in ("pblh", "tpert", "landfrac", "cld")]
outs = [mid(), mid(), mid(), mid(), itf(), col(), col(), col(), col(), col(), col(), mid(), mid(), mid(), mid(), itf(), itf(), itf(),
        mid(), mid(), mid(), mid(), mid(), mid(), col()]
ints = [np.zeros(nch * 16, np.int32) for _ in range(4)]
tail = [col(), col(), col(), col(), np.zeros(2)]
f.lib.hfd_zm_conv_tend(C.c_int(n), C.c_double(synth.ZTODT), *[a.ctypes.data_as(D) for a in ins], *[a.ctypes.data_as(D) for a in outs],
                       *[a.ctypes.data_as(I) for a in ints], *[a.ctypes.data_as(D) for a in tail])
print("RETURNED deep", int(ints[3][:nch].sum()))
"""


@pytest.mark.parametrize("case,halts", [("clean", False), ("nan", True)])
def test_fortran_whole_rank_dispatch_halts_where_the_reference_halts(case, halts):
    """colphys_rank_run_zm_conv_tend reads the status words of the whole rank: a sounding that is not a number ends the run
    with the reference's ienthalpy message (zm_conv.F90:5565; endrun here is the stand-in that stops the process, hence the
    child process); clean soundings return."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _HF_HALT, ROOT, case], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    if halts:
        assert r.returncode != 0 and "RETURNED" not in r.stdout, out[-2000:]
        assert "ENDRUN:" in out and "**** ZM_CONV IENTHALPY: Tmix did not converge ****" in out, out[-2000:]
    else:
        assert r.returncode == 0 and "RETURNED deep" in r.stdout and "ENDRUN:" not in out, out[-2000:]
