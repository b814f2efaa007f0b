"""Seeded synthetic physics_state soundings and aerosol inputs.

The reference ships no input data, so every parity test, golden fixture and bench run
uses columns drawn here.  Layout of every profile array is "column-contiguous": numpy [ncol][pver]
(interfaces [ncol][pver+1]), which is what the C-ABI's device entry points take (include/colphys.h).

Field names follow physics_state (physics/physics_types.F90:62-121) and the zm_convr / zm_conv_evap /
momtran dummy arguments (physics/zm_conv.F90:231-244,1712-1717,2315-2319).
"""
from __future__ import annotations

import numpy as np

# CESM share-code constants (shr_const_mod), the values CAM's physconst hands to zm_convi.
BOLTZ = 1.38065e-23
AVOGAD = 6.02214e26
RGAS_U = AVOGAD * BOLTZ
MWDAIR = 28.966
MWWV = 18.016
PHYSCONST = dict(
    cpair=1.00464e3,
    epsilo=MWWV / MWDAIR,
    gravit=9.80616,
    latice=3.337e5,
    latvap=2.501e6,
    tmelt=273.15,
    rair=RGAS_U / MWDAIR,
    cpwv=1.810e3,
    cpliq=4.188e3,
    rh2o=RGAS_U / MWWV,
    rhoh2o=1.000e3,
)
PHYSCONST["zvir"] = PHYSCONST["rh2o"] / PHYSCONST["rair"] - 1.0
PHYSCONST["cpvir"] = PHYSCONST["cpwv"] / PHYSCONST["cpair"] - 1.0
PHYSCONST["rga"] = 1.0 / PHYSCONST["gravit"]

# zmconv_nl values used throughout; ztodt = 1800 s => delt = 900 s
ZM_NAMELIST = dict(
    c0_lnd=0.0075, c0_ocn=0.03, ke=5.0e-6, ke_lnd=1.0e-5, momcu=0.7, momcd=0.7, num_cin=1,
    org=0, microp=0, no_deep_pbl=0, tiedke_add=0.5, capelmt=70.0, dmpdz=-1.0e-3, parcel_pbl=0, tau=3600.0,
)
ZTODT = 1800.0
PTOP = 226.0
DEFAULT_SEED = 20261004


def sigma_edges(pver: int) -> np.ndarray:
    k = np.arange(pver + 1, dtype=np.float64)
    return (k / pver) ** 1.6


def limcnv_from_pref(pver: int, ps_ref: float = 1.0e5) -> int:
    """First interface at or below 40 hPa (zm_conv_intr.F90:356-368), 1-based."""
    pref_edge = PTOP + sigma_edges(pver) * (ps_ref - PTOP)
    if pref_edge[0] >= 4.0e3:
        return 1
    for k in range(pver):
        if pref_edge[k] < 4.0e3 <= pref_edge[k + 1]:
            return k + 1
    return pver + 1


def _svp_water(t):
    tboil = 373.16
    return 10.0 ** (-7.90298 * (tboil / t - 1.0) + 5.02808 * np.log10(tboil / t)
                    - 1.3816e-7 * (10.0 ** (11.344 * (1.0 - t / tboil)) - 1.0)
                    + 8.1328e-3 * (10.0 ** (-3.49149 * (tboil / t - 1.0)) - 1.0)
                    + np.log10(1013.246)) * 100.0


def geopotential_fv(state: dict) -> tuple[np.ndarray, np.ndarray]:
    """numpy statement of the FV branch of geopotential_t, only to give the generator consistent zi/zm.
    (The tested implementations are the oracle and the HIP kernel, never this.)"""
    c = PHYSCONST
    ncol, pver = state["t"].shape
    zi = np.zeros((ncol, pver + 1))
    zm = np.zeros((ncol, pver))
    rog = c["rair"] / c["gravit"]
    for k in range(pver - 1, -1, -1):
        hkl = state["lnpint"][:, k + 1] - state["lnpint"][:, k]
        hkk = 1.0 - state["pint"][:, k] * hkl * state["rpdel"][:, k]
        tv = state["t"][:, k] * (1.0 + c["zvir"] * state["q"][:, k])
        zm[:, k] = zi[:, k + 1] + rog * tv * hkk
        zi[:, k] = zi[:, k + 1] + rog * tv * hkl
    return zi, zm


def make_state(ncol: int, pver: int, seed: int = DEFAULT_SEED, *, moist_boost: float = 1.0,
               ts_range=(275.0, 303.0), rh_range=(0.3, 0.95)) -> dict:
    """Tropical-ish soundings; 10-30 % of them trigger deep convection with ZM_NAMELIST."""
    c = PHYSCONST
    rng = np.random.Generator(np.random.PCG64(seed))
    ps = rng.uniform(95000.0, 103000.0, ncol)
    ts = rng.uniform(ts_range[0], ts_range[1], ncol)
    gam = rng.uniform(5.5e-3, 7.0e-3, ncol)
    rh = rng.uniform(rh_range[0], rh_range[1], ncol)
    sig = sigma_edges(pver)
    pint = PTOP + sig[None, :] * (ps[:, None] - PTOP)
    pmid = 0.5 * (pint[:, :-1] + pint[:, 1:])
    pdel = pint[:, 1:] - pint[:, :-1]
    t = np.maximum(195.0, ts[:, None] * (pmid / ps[:, None]) ** (c["rair"] * gam[:, None] / c["gravit"]))
    es = _svp_water(t)
    qs = c["epsilo"] * es / np.maximum(pmid - (1.0 - c["epsilo"]) * es, 1.0)
    q = np.clip(moist_boost * rh[:, None] * qs * (pmid / ps[:, None]) ** 1.5, 1.0e-6, 0.03)
    shear = rng.uniform(-20.0, 20.0, (ncol, 2))
    base = rng.uniform(-8.0, 8.0, (ncol, 2))
    hgt = 1.0 - pmid / ps[:, None]
    u = base[:, 0:1] + shear[:, 0:1] * hgt + rng.normal(0.0, 0.5, (ncol, pver))
    v = base[:, 1:2] + shear[:, 1:2] * hgt + rng.normal(0.0, 0.5, (ncol, pver))
    state = dict(
        ncol=ncol, pver=pver,
        ps=ps, t=np.ascontiguousarray(t), q=np.ascontiguousarray(q), u=u, v=v,
        pint=np.ascontiguousarray(pint), pmid=np.ascontiguousarray(pmid), pdel=np.ascontiguousarray(pdel),
        rpdel=1.0 / pdel, lnpint=np.log(pint), lnpmid=np.log(pmid),
        pdeldry=np.ascontiguousarray(pdel * (1.0 - q)),
        phis=np.where(rng.uniform(size=ncol) < 0.7, 0.0, rng.uniform(0.0, 2.0e4, ncol)),
        pblh=rng.uniform(300.0, 1500.0, ncol),
        tpert=rng.uniform(0.0, 1.0, ncol),
        landfrac=(rng.uniform(size=ncol) < 0.3).astype(np.float64),
        cld=rng.uniform(0.0, 0.3, (ncol, pver)),
        rair=np.full((ncol, pver), c["rair"]),
        zvir=np.full((ncol, pver), c["zvir"]),
    )
    state["zi"], state["zm"] = geopotential_fv(state)
    return state


# ---------------------------------------------------------------------------------------------------
# aerosol optics inputs (modal_aer_opt.F90:485-1469); MAM4-like: 4 modes, species {6,4,3,2}
# ---------------------------------------------------------------------------------------------------
NSWBANDS = 14
NLWBANDS = 16
NCOEF, PREFR, PREFI = 5, 7, 10
IDX_SW_DIAG, IDX_NIR_DIAG, IDX_UV_DIAG = 10, 8, 11     # radconstants, 1-based
SPEC_TYPES = ["sulfate", "p-organic", "s-organic", "black-c", "dust", "seasalt"]
NSPEC_M = (6, 4, 3, 2)


# a MAM7-like variant for the tests: seven modes, up to eight species, modes that hold two species of one type (the species
# partition of modal_aero_sw keeps the LAST one, modal_aer_opt.F90:780-829) and species of no partition type (-1)
NSPEC_M7 = (6, 4, 7, 3, 3, 8, 2)
SPECTYPE_M7 = ((0, 1, 2, 3, 4, 5, -1, -1), (0, 2, 5, 0, -1, -1, -1, -1), (4, 4, 0, 5, 3, 1, 4, -1), (4, 5, 0, -1, -1, -1, -1, -1),
               (-1, 0, 5, -1, -1, -1, -1, -1), (3, 3, 1, 1, 2, 0, 5, 4), (0, -1, -1, -1, -1, -1, -1, -1))


def make_aero_props(seed: int = DEFAULT_SEED, nspec_m=NSPEC_M, spectype=None) -> dict:
    """Per-mode / per-species property tables: smooth, seeded, physically plausible magnitudes.
    nspec_m / spectype: species counts per mode and their types [mode][species] (default: the MAM4-like set, types cycling)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    nm, nsmax = len(nspec_m), max(nspec_m)
    props = dict(nmodes=nm, nspec=np.array(nspec_m, dtype=np.int32), nspec_max=nsmax)
    props["sigmag"] = np.array([1.8, 1.6] * nm)[:nm]
    props["specdens"] = rng.uniform(1000.0, 2600.0, (nm, nsmax))
    props["hygro"] = rng.uniform(1.0e-10, 1.2, (nm, nsmax))
    if spectype is None:
        props["spectype"] = np.array([[(l + m) % 6 for l in range(nsmax)] for m in range(nm)], dtype=np.int32)
    else:
        props["spectype"] = np.array([list(row)[:nsmax] for row in spectype], dtype=np.int32)
    props["specrefindex_sw"] = (rng.uniform(1.3, 1.9, (nm, nsmax, NSWBANDS))
                                + 1j * rng.uniform(1.0e-9, 0.4, (nm, nsmax, NSWBANDS)))
    props["specrefindex_lw"] = (rng.uniform(1.2, 2.4, (nm, nsmax, NLWBANDS))
                                + 1j * rng.uniform(1.0e-3, 0.8, (nm, nsmax, NLWBANDS)))
    props["crefwsw"] = rng.uniform(1.30, 1.36, NSWBANDS) + 1j * np.abs(rng.uniform(1.0e-9, 0.05, NSWBANDS))
    props["crefwlw"] = rng.uniform(1.15, 1.6, NLWBANDS) + 1j * np.abs(rng.uniform(0.01, 0.5, NLWBANDS))
    # refractive-index grids per (mode, band): ascending; the synthetic mixtures fall inside and outside them
    refr0 = np.linspace(1.32, 1.85, PREFR)
    refi0 = np.concatenate([[0.0], np.geomspace(1.0e-4, 0.5, PREFI - 1)])
    props["refrtabsw"] = np.tile(refr0, (nm, NSWBANDS, 1)) * (1.0 + 0.01 * rng.uniform(-1, 1, (nm, NSWBANDS, 1)))
    props["refitabsw"] = np.tile(refi0, (nm, NSWBANDS, 1)) * (1.0 + 0.05 * rng.uniform(-1, 1, (nm, NSWBANDS, 1)))
    props["refrtablw"] = np.tile(np.linspace(1.2, 2.3, PREFR), (nm, NLWBANDS, 1))
    props["refitablw"] = np.tile(np.concatenate([[0.0], np.geomspace(1.0e-3, 0.9, PREFI - 1)]), (nm, NLWBANDS, 1))

    # Chebyshev coefficient tables [mode][band][prefi][prefr][ncoef]  (Fortran (ncoef,prefr,prefi,band))
    def cheb_table(nb, c0, spread):
        ir = np.linspace(0.0, 1.0, PREFR)[None, None, None, :, None]
        ii = np.linspace(0.0, 1.0, PREFI)[None, None, :, None, None]
        nc = np.arange(NCOEF)[None, None, None, None, :]
        base = c0 * (nc == 0) + spread * 0.5 ** nc * np.cos(1.3 * nc + 2.0 * ir + 1.1 * ii)
        return base + 0.05 * spread * rng.standard_normal((nm, nb, PREFI, PREFR, NCOEF))

    props["extpsw"] = cheb_table(NSWBANDS, 2.0 * np.log(3.0e3), 0.8)      # exp(0.5*c1+...) ~ m2/kg
    props["abspsw"] = cheb_table(NSWBANDS, 2.0 * 2.0e2, 60.0)
    props["asmpsw"] = cheb_table(NSWBANDS, 2.0 * 0.65, 0.08)
    props["absplw"] = cheb_table(NLWBANDS, 2.0 * 1.5e2, 50.0)
    return props


def make_aero_state(ncol: int, pver: int, props: dict, seed: int = DEFAULT_SEED) -> dict:
    rng = np.random.Generator(np.random.PCG64(seed + 11))
    nm, nsmax = props["nmodes"], props["nspec_max"]

    def logu(lo, hi, shape):
        return np.exp(rng.uniform(np.log(lo), np.log(hi), shape))

    aer = dict(
        specmmr=logu(1.0e-12, 1.0e-8, (nm, nsmax, ncol, pver)),
        dgnumwet=logu(2.0e-8, 5.0e-6, (nm, ncol, pver)),
        qaerwat=logu(1.0e-12, 1.0e-7, (nm, ncol, pver)),
        troplevchem=rng.integers(max(2, pver // 4), max(3, pver // 2), ncol).astype(np.int32),
    )
    # a few coarse particles beyond rmmax=25 um exercise the geometric-optics branch (modal_aer_opt.F90:869-877)
    big = rng.uniform(size=(nm, ncol, pver)) < 0.01
    aer["dgnumwet"][big] = rng.uniform(2.0e-5, 6.0e-5, big.sum())
    # a few slightly negative water contents exercise the watervol<0 reset (modal_aer_opt.F90:837-844)
    neg = rng.uniform(size=(nm, ncol, pver)) < 0.005
    aer["qaerwat"][neg] = -1.0e-16
    for m in range(nm):                       # unused species slots hold zeros
        aer["specmmr"][m, props["nspec"][m]:] = 0.0
    return aer
