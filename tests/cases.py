"""Input builders shared by the parity tests (CPU oracle side and GPU side)."""
from __future__ import annotations

import os

import numpy as np

from cam_nor_physics_amd import abi, synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ZM_GOLDEN = ["zm_l32", "zm_l70", "zm_l32_cold", "zm_l32_pbl"]
AERO_GOLDEN = ["aero_l32", "aero_l70", "aero_m7"]
PCOLS = 16


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: g[k] for k in g.files}


def golden_namelist(g):
    nl = {k: float(v) for k, v in zip(g["namelist_keys"], g["namelist_vals"])}
    for k in ("num_cin", "org", "microp", "no_deep_pbl", "parcel_pbl"):
        nl[k] = int(nl[k])
    return nl


def golden_params(g):
    return abi.make_zm_params(synth.PHYSCONST, golden_namelist(g), int(g["limcnv"]))


def default_params(pver):
    return abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, synth.limcnv_from_pref(pver))


def zm_args_from_golden(g):
    """positional args of zm_convr(t, qh, pblh, zm, geos, zi, pap, paph, dpp, delt, tpert, landfrac)"""
    return (g["in_t"], g["in_q"], g["in_pblh"], g["geo_fv_zm"], g["in_phis"], g["geo_fv_zi"], g["in_pmid"], g["in_pint"],
            g["in_pdel"], 0.5 * synth.ZTODT, g["in_tpert"], g["in_landfrac"])


def zm_args_from_state(st, zi, zm):
    return (st["t"], st["q"], st["pblh"], zm, st["phis"], zi, st["pmid"], st["pint"], st["pdel"], 0.5 * synth.ZTODT,
            st["tpert"], st["landfrac"])


def geo_args(st_or_g, prefix=""):
    s = st_or_g
    k = (lambda n: s[prefix + n])
    return (k("lnpint"), k("lnpmid"), k("pint"), k("pmid"), k("pdel"), k("rpdel"), k("t"))


def evap_cold_case(ncol, pver, seed):
    """zm_conv_evap inputs that exercise snow production, melting and surface snowfall: cold soundings
    with a synthetic precipitation-production profile (negative in places, as downdraft evaporation gives)."""
    st = synth.make_state(ncol, pver, seed, ts_range=(248.0, 284.0), rh_range=(0.4, 0.98))
    rng = np.random.default_rng(seed + 3)
    prd = rng.uniform(-2.0e-9, 4.0e-8, (ncol, pver))
    prd[:, : pver // 4] = 0.0
    prd[rng.uniform(size=ncol) < 0.2] = 0.0
    prec = np.maximum((prd * st["pdel"]).sum(axis=1) / synth.PHYSCONST["gravit"] / 1000.0, 0.0)
    return (st["t"], st["pmid"], st["pdel"], st["q"], st["landfrac"], prd, st["cld"], synth.ZTODT, prec)


NCNST = 6
DOCONVTRAN = np.array([0, 1, 1, 0, 1, 1], dtype=np.int32)     # tracer 1 (water vapour) is never transported
ISDRY = np.array([0, 0, 1, 0, 1, 0], dtype=np.int32)


def convtran_inputs(pdel, qv, ideep, lengath, seed, pcols=PCOLS):
    """tracer fields [NCNST][ncol][pver], insoluble fractions, and the gathered dry pressure thickness in mb
    (zm_conv_intr.F90:1010-1017: dpdry(i,:) = state%pdeldry(ideep(i),:)/100)."""
    ncol, pver = pdel.shape
    rng = np.random.default_rng(seed)
    lev = np.linspace(0.0, 1.0, pver)[None, None, :]
    q = rng.uniform(0.2, 1.0, (NCNST, ncol, 1)) * 1.0e-6 * (0.05 + lev ** rng.uniform(0.5, 3.0, (NCNST, ncol, 1)))
    q *= 1.0 + 0.2 * rng.standard_normal((NCNST, ncol, pver))
    q[2, :, : pver // 3] = 0.0                                  # exact zeros: the small-difference branch
    q[4, ::7, pver // 2] *= -1.0                                # a few negative values: minc < 0 branch
    fracis = rng.uniform(0.0, 1.0, (NCNST, ncol, pver))
    pdeldry = pdel * (1.0 - qv)
    dpdry = np.zeros((ncol, pver))
    for c in range(len(lengath)):
        for j in range(int(lengath[c])):
            row = c * pcols + j
            dpdry[row] = pdeldry[c * pcols + int(ideep[row]) - 1] / 100.0
    dpdry[dpdry == 0.0] = 1.0                                   # slots beyond lengath are never read; keep them finite
    return np.ascontiguousarray(q), np.ascontiguousarray(fracis), dpdry


def geo_dse_case(ncol, pver, seed, dycore, geo_oracle):
    """Arguments of geopotential_dse for one dycore branch: dse = cpair*T + gravit*zm + phis with zm from `geo_oracle`'s
    geopotential_t (any of the oracles: the dse array it yields is stored with the golden, so every implementation is
    tested on the same bits), cpair varying by 2 % across columns and levels."""
    st = synth.make_state(ncol, pver, seed)
    c = synth.PHYSCONST
    rng = np.random.default_rng(seed + 5)
    cpair = c["cpair"] * (1.0 + 0.02 * rng.uniform(-1.0, 1.0, (ncol, pver)))
    nsp = 2 if dycore == 2 else 1
    q = np.stack([st["q"], 0.1 * st["q"]])
    _, zm = geo_oracle.geopotential_t(st["lnpint"], st["lnpmid"], st["pint"], st["pmid"], st["pdel"], st["rpdel"], st["t"], q,
                                      st["rair"], c["gravit"], st["zvir"], dycore=dycore, nspecies=nsp)
    dse = cpair * st["t"] + c["gravit"] * zm + st["phis"][:, None]
    args = (st["lnpint"], st["lnpmid"], st["pint"], st["pmid"], st["pdel"], st["rpdel"], dse, q, st["phis"], st["rair"],
            c["gravit"], cpair, st["zvir"])
    return dict(args=args, nspecies=nsp, dse=dse, t=st["t"])


def geo_dse_args_from_golden(g, tag, ncol, pver, seed, dycore):
    """The same arguments with the golden's stored dse (bit-identical inputs for every implementation)."""
    st = synth.make_state(ncol, pver, seed)
    c = synth.PHYSCONST
    rng = np.random.default_rng(seed + 5)
    cpair = c["cpair"] * (1.0 + 0.02 * rng.uniform(-1.0, 1.0, (ncol, pver)))
    q = np.stack([st["q"], 0.1 * st["q"]])
    return (st["lnpint"], st["lnpmid"], st["pint"], st["pmid"], st["pdel"], st["rpdel"], g[f"{tag}_dse"], q, st["phis"], st["rair"],
            c["gravit"], cpair, st["zvir"]), (2 if dycore == 2 else 1), st["t"]


def convect_diag_case(ncol, pver, seed):
    """Inputs of convect_diagnostics_calc (convect_diagnostics.F90:115-249): deep outputs shaped like zm_conv_tend's
    (cloud top / base as real-valued level indices, pver / 1 where nothing convects) and a synthetic shallow scheme."""
    rng = np.random.default_rng(seed)
    st = synth.make_state(ncol, pver, seed)
    deep = rng.uniform(size=ncol) < 0.3
    cnt = np.where(deep, rng.integers(pver // 3, pver - 3, ncol), pver).astype(np.float64)
    cnb = np.where(deep, rng.integers(pver - 3, pver + 1, ncol), 1).astype(np.float64)
    sh = rng.uniform(size=ncol) < 0.5
    f = dict(
        cmfmc=rng.uniform(0, 0.05, (ncol, pver + 1)) * deep[:, None], qc=rng.uniform(0, 1e-8, (ncol, pver)) * deep[:, None],
        rliq=rng.uniform(0, 1e-8, ncol) * deep, cnt=cnt, cnb=cnb,
        cmfmc2=rng.uniform(0, 0.02, (ncol, pver + 1)) * sh[:, None], qc2=rng.uniform(0, 1e-8, (ncol, pver)) * sh[:, None],
        rprdsh=rng.uniform(0, 1e-8, (ncol, pver)), rliq2=rng.uniform(0, 1e-8, ncol) * sh,
        cnt2=np.where(sh, rng.integers(pver // 2, pver - 1, ncol), pver).astype(np.float64),
        cnb2=np.where(sh, rng.integers(pver - 2, pver + 1, ncol), 1).astype(np.float64),
        precc=rng.uniform(0, 1e-7, ncol), snow=rng.uniform(0, 1e-8, ncol), icwmr=rng.uniform(0, 1e-4, (ncol, pver)),
        evapcsh=rng.uniform(0, 1e-8, (ncol, pver)), rprddp=rng.uniform(0, 1e-8, (ncol, pver)) * deep[:, None], pmid=st["pmid"],
        rprdtot=np.full((ncol, pver), np.nan), pcnt=np.full(ncol, np.nan), pcnb=np.full(ncol, np.nan))
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in f.items()}


def convect_diag_numpy(f, clubb):
    """Independent restatement (array form) of convect_diagnostics.F90:178-231"""
    o = {k: v.copy() for k, v in f.items()}
    pver = o["qc"].shape[1]
    if clubb:
        for k in ("cmfmc2", "rprdsh", "precc", "icwmr", "rliq2", "qc2", "evapcsh", "snow"):
            o[k][...] = 0.0
        cnt2, cnb2 = np.full_like(o["cnt"], pver), np.ones_like(o["cnb"])
    else:
        cnt2, cnb2 = o["cnt2"], o["cnb2"]
    o["cmfmc"] = o["cmfmc"] + o["cmfmc2"]
    cnt = np.where(cnt2 < o["cnt"], cnt2, o["cnt"])
    cnb = np.where(cnb2 > o["cnb"], cnb2, o["cnb"])
    cnb = np.where(cnb == 1.0, cnt, cnb)
    idx = np.arange(len(cnt))
    o["cnt"], o["cnb"] = cnt, cnb
    o["pcnt"] = o["pmid"][idx, cnt.astype(int) - 1]
    o["pcnb"] = o["pmid"][idx, cnb.astype(int) - 1]
    o["rprdtot"] = o["rprdsh"] + o["rprddp"]
    o["qc"] = o["qc"] + o["qc2"]
    o["rliq"] = o["rliq"] + o["rliq2"]
    return o


def derived_fields_case(ncol, pver, ncnst, seed):
    """Inputs of d_p_coupling's derived-fields block: interface pressures, T, constituents with some negative bottom-layer
    values (both outcomes of the fix: payable by the level above or not), constituent-dependent cpairv / rairv."""
    st = synth.make_state(ncol, pver, seed)
    c = synth.PHYSCONST
    rng = np.random.default_rng(seed + 17)
    q = np.empty((ncnst, ncol, pver))
    q[0] = st["q"]
    for m in range(1, ncnst):
        q[m] = rng.uniform(0.0, 1.0e-6, (ncol, pver))
    neg = rng.uniform(size=(ncnst, ncol)) < 0.15
    q[:, :, -1] = np.where(neg, -rng.uniform(0.0, 2.0e-7, (ncnst, ncol)), q[:, :, -1])
    poor = rng.uniform(size=(ncnst, ncol)) < 0.3                       # level above cannot pay
    q[:, :, -2] = np.where(neg & poor, 1.0e-12, q[:, :, -2])
    qmin = np.full(ncnst, 0.0); qmin[0] = 1.0e-12
    cpair = c["cpair"] * (1.0 + 0.01 * rng.uniform(-1, 1, (ncol, pver)))
    return dict(pint=st["pint"], lnpint=st["lnpint"], t=st["t"], q=np.ascontiguousarray(q), qmin=qmin, phis=st["phis"], cpair=cpair,
                rair=st["rair"], zvir=st["zvir"], gravit=c["gravit"])


def derived_fields_numpy(a, geo_oracle, dycore, nspecies):
    """Independent array-form restatement of fv/dp_coupling.F90:552-559, 565-590, 617-629 (geopotential_t from `geo_oracle`)"""
    pint, q = a["pint"], a["q"].copy()
    pdel = pint[:, 1:] - pint[:, :-1]
    rpdel = 1.0 / pdel
    pmid = 0.5 * (pint[:, :-1] + pint[:, 1:])
    lnpmid = np.log(pmid)
    for m in range(q.shape[0]):
        bot, up = q[m, :, -1], q[m, :, -2]
        need = bot < a["qmin"][m]
        dqreq = (a["qmin"][m] - bot) * pdel[:, -1] / pdel[:, -2]
        ok = need & (dqreq < up - a["qmin"][m])
        q[m, :, -2] = np.where(ok, up - dqreq, up)
        q[m, :, -1] = np.where(ok, a["qmin"][m], bot)
    zi, zm = geo_oracle.geopotential_t(a["lnpint"], lnpmid, pint, pmid, pdel, rpdel, a["t"], q, a["rair"], a["gravit"], a["zvir"],
                                       dycore=dycore, nspecies=nspecies)
    s = a["cpair"] * a["t"] + a["gravit"] * zm + a["phis"][:, None]
    return dict(pdel=pdel, rpdel=rpdel, pmid=pmid, lnpmid=lnpmid, zi=zi, zm=zm, s=s, q=q)


# ---- modal_aero_sw history diagnostics -------------------------------------------------------------------------
SWD_FILL = 1.0e36          # cam_history_support's fillvalue


def swd_reference_ssavis(col, nite=None):
    """The caller-side last step of SSAVISdn (modal_aer_opt.F90:1102-1128): the engine / the C restatement return the sum
    of dopaer*palb; the reference divides it by aodvis AFTER aodvis' night columns have been overwritten with fillvalue.
    Returns a copy of the packed column fields with that row finished the reference's way."""
    from cam_nor_physics_amd import abi
    col = np.array(col, dtype=np.float64, copy=True)
    i_vis, i_ssa = abi.SWD_COL_FIELDS.index("AODVISdn"), abi.SWD_COL_FIELDS.index("SSAVISdn")
    aodvis = col[i_vis].copy()
    if nite is not None:
        aodvis[np.asarray(nite) != 0] = SWD_FILL
    col[i_ssa] = np.where(aodvis > 1.0e-10, col[i_ssa] / aodvis, 0.925)
    return col


def swd_field_names(nmodes):
    from cam_nor_physics_amd import abi
    names = list(abi.SWD_COL_FIELDS)
    for m in range(1, nmodes + 1):
        names += [f"BURDENdn{m:02d}", f"AODdn_mode{m}", f"AODdnDUST{m:02d}"]
    return list(abi.SWD_LEV_FIELDS), names


def aero_props_from_golden(g):
    """the property tables an optics fixture was generated with (tests/golden/make_golden.py aero_golden)"""
    from cam_nor_physics_amd import synth
    if "m7" in g and int(g["m7"]):
        return synth.make_aero_props(int(g["seed"]), synth.NSPEC_M7, synth.SPECTYPE_M7)
    return synth.make_aero_props(int(g["seed"]))


# ---- the operator layer: zm_conv_tend / zm_conv_tend_2 (zm_conv_intr.F90:390-1028) --------------------------------
INTR_GOLDEN = ["intr_l32", "intr_l70"]


def intr_state_from_golden(g):
    st = {k[3:]: v for k, v in g.items() if k.startswith("in_") and k not in ("in_q6", "in_fracis6")}
    st["q"] = g["in_q6"][0]
    st["ncol"] = int(g["ncol"])
    return st


def zm_conv_tend_composed(o, st, q6, fracis6, lq1, isdry, dt, qmin=1.0e-12, pcols=PCOLS):
    """zm_conv_tend (zm_conv_intr.F90:441-905) restated as a composition of the column routines of `o` (an oracle object:
    zm_convr, zm_conv_evap, momtran, convtran, geopotential_t) with physics_update / physics_ptend_sum (physics_types.F90:292-340,
    425-447, 698-844) in numpy between them.  Returns ptend_all, the pbuf fields and the outfld values under the golden's names."""
    c = synth.PHYSCONST
    cp, g_ = c["cpair"], c["gravit"]
    n, pver = st["t"].shape
    r = {}
    z = o.zm_convr(st["t"], st["q"], st["pblh"], st["zm"], st["phis"], st["zi"], st["pmid"], st["pint"], st["pdel"], 0.5 * dt,
                   st["tpert"], st["landfrac"])                                              # :662-675
    for k in ("ql", "rprd", "dlf", "dif", "mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg", "ideep", "lengath", "rliq", "rice",
              "jctop", "jcbot", "cme", "zdu", "pflx"):
        r[k] = z[k]
    r["mcon"] = z["mcon"] * 100.0 / g_                                                        # :691-692
    r["mconzm"] = r["mcon"]
    h = {"CAPE": z["cape"], "EURT": z["eurt"][:, pver - 1], "CMFMC_DP": r["mcon"], "ZMDT": z["heat"] / cp, "ZMDQ": z["qtnd"],
         "DIFZM": z["dif"], "DLFZM": z["dlf"]}
    freq = np.zeros(n); mu_out = np.zeros((n, pver)); md_out = np.zeros((n, pver))
    ps = st["pint"][:, -1]
    pcont, pconb = ps.copy(), ps.copy()
    for ic, lg in enumerate(z["lengath"]):                                                   # :685-688, 700-706, 722-729
        for j in range(int(lg)):
            row = ic * pcols + j
            ii = ic * pcols + int(z["ideep"][row]) - 1
            freq[ii] = 1.0
            mu_out[ii] = z["mu"][row] * 100.0 / g_
            md_out[ii] = z["md"][row] * 100.0 / g_
            if z["maxg"][row] > z["jt"][row]:
                pcont[ii] = st["pmid"][ii, z["jt"][row] - 1]
                pconb[ii] = st["pmid"][ii, z["maxg"][row] - 1]
    h.update(FREQZM=freq, ZMMU=mu_out, ZMMD=md_out, PCONVT=pcont, PCONVB=pconb)
    # physics_update #1 (:737-739)
    t1 = st["t"] + z["heat"] * dt / cp
    q1 = np.maximum(st["q"] + z["qtnd"] * dt, qmin)
    e = o.zm_conv_evap(t1, st["pmid"], st["pdel"], q1, st["landfrac"], z["rprd"], st["cld"], dt, z["prec"])      # :764-769
    r.update(prec=e["prec"], snow=e["snow"], evapcdp=e["tend_q"], flxprec=e["flxprec"], flxsnow=e["flxsnow"])
    h.update(EVAPTZM=e["tend_s"] / cp, FZSNTZM=e["tend_s_snwprd"] / cp, EVSNTZM=e["tend_s_snwevmlt"] / cp, EVAPQZM=e["tend_q"],
             ZMFLXPRC=e["flxprec"], ZMFLXSNW=e["flxsnow"], ZMNTPRPD=e["ntprprd"], ZMNTSNPD=e["ntsnprd"], ZMEIHEAT=e["tend_s"],
             PRECCDZM=e["prec"], PRECZ=e["prec"])
    t2 = t1 + e["tend_s"] * dt / cp                                                           # :802-806
    q2 = np.maximum(q1 + e["tend_q"] * dt, qmin)
    m = o.momtran(st["u"], st["v"], z["mu"], z["md"], z["du"], z["eu"], z["ed"], z["dp"], z["dsubcld"], z["jt"], z["maxg"], z["ideep"],
                  z["lengath"], dt)                                                          # :822-826 (winds = state1 u, v = state u, v)
    h.update(ZMMTT=m["seten"] / cp, ZMMTU=m["dqdt"][0], ZMMTV=m["dqdt"][1], ZMUPGU=m["pguall"][0], ZMUPGD=m["pgdall"][0],
             ZMVPGU=m["pguall"][1], ZMVPGD=m["pgdall"][1], ZMICUU=m["icwu"][0], ZMICUD=m["icwd"][0], ZMICVU=m["icwu"][1], ZMICVD=m["icwd"][1])
    r["ptend_s"] = (z["heat"] + e["tend_s"]) + m["seten"]                                     # physics_ptend_sum x3
    r["ptend_u"], r["ptend_v"] = m["dqdt"][0], m["dqdt"][1]
    # convtran1 (:860-884): cloud liquid and ice of state1 (untouched by the three updates), moist mixing ratios, dpdry = 0
    ncn = q6.shape[0]
    do = np.zeros(ncn, np.int32); do[:] = np.asarray(lq1)[:ncn]
    ct = o.convtran(q6, fracis6, z["mu"], z["md"], z["du"], z["eu"], z["ed"], z["dp"], np.ones_like(z["dp"]), z["jt"], z["maxg"], z["ideep"],
                    z["lengath"], do, np.zeros(ncn, np.int32), dt)
    pq = np.zeros_like(q6)
    pq[0] = z["qtnd"] + e["tend_q"]
    for mm in range(1, ncn):
        if do[mm]:
            pq[mm] = ct[mm]
    r["ptend_q"] = pq
    h.update(ZMDLIQ=ct[1], ZMDICE=ct[2])
    r["state1"] = dict(t=t2 + m["seten"] * dt / cp, q=q2, u=st["u"] + m["dqdt"][0] * dt, v=st["v"] + m["dqdt"][1] * dt)
    r["hist"] = h
    return r


def assert_tend_matches_golden(r, g, pcols=PCOLS, hist=True):
    """r: zm_conv_tend_composed's dictionary (or an engine's outputs under the same names); g: an intr_* golden"""
    from util import assert_close, gathered_mask
    n = int(g["ncol"])
    assert np.array_equal(r["lengath"], g["tend_lengath"])
    m = gathered_mask(g["tend_lengath"], n, pcols)
    for k in ("mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg", "ideep"):
        if k in r:
            assert_close("zm_conv_tend." + k, r[k][m], g["tend_" + k][m])
    for k in ("ptend_s", "ptend_u", "ptend_v", "mcon", "pflx", "cme", "zdu", "rliq", "rice", "jctop", "jcbot", "prec", "snow", "ql", "rprd",
              "evapcdp", "dlf", "dif", "mconzm", "flxprec", "flxsnow"):
        if k in r:
            assert_close("zm_conv_tend." + k, r[k], g["tend_" + k])
    if "ptend_q" in r:
        for mm in range(r["ptend_q"].shape[0]):
            assert bool(g["tend_ptend_lq"][mm]) == (mm < 3), mm
            assert_close(f"zm_conv_tend.ptend_q[{mm}]", r["ptend_q"][mm], g["tend_ptend_q"][mm])
    if hist:
        for k, v in r["hist"].items():
            assert_close("outfld " + k, v, g["hist_" + k])


def dry_tail_numpy(pint, pdel, q, isdry, qmin):
    """array-form restatement of set_state_pdry + set_wet_to_dry (physics_types.F90:1925-1984) + qneg3 as d_p_coupling calls them
    (fv/dp_coupling.F90:636-651); q [ncnst][ncol][pver], water vapour first"""
    n, pver = pdel.shape
    pdeldry = pdel * (1.0 - q[0])
    pintdry = np.empty((n, pver + 1)); pintdry[:, 0] = pint[:, 0]
    psdry = pint[:, 0].copy()
    pmiddry = np.empty((n, pver))
    for k in range(pver):
        pintdry[:, k + 1] = pintdry[:, k] + pdeldry[:, k]
        pmiddry[:, k] = (pintdry[:, k + 1] + pintdry[:, k]) / 2.0
        psdry = psdry + pdeldry[:, k]
    qo = q.copy()
    for m in range(q.shape[0]):
        if isdry[m]:
            qo[m] = q[m] * pdel / pdeldry
        qo[m] = np.where(qo[m] < qmin[m], qmin[m], qo[m])
    return dict(pdeldry=pdeldry, rpdeldry=1.0 / pdeldry, pmiddry=pmiddry, lnpmiddry=np.log(pmiddry), pintdry=pintdry,
                lnpintdry=np.log(pintdry), psdry=psdry, q=qo)
