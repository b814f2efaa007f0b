"""ctypes mirror of include/colphys.h (the C ABI).  Struct layouts must match the header field for field."""
from __future__ import annotations

import ctypes as C

import numpy as np

PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int32)

COLPHYS_MAX_PVER = 128
DYCORE_EUL, DYCORE_FV, DYCORE_SE = 0, 1, 2
STATUS_BRENT_NOT_CONVERGED = 1
STATUS_DOPAER_RANGE = 2
STATUS_DOPAER_NEGATIVE = 4
STATUS_BRENT_ENTHALPY = 8
STATUS_COUNT_SHIFT = 8
NCOEF, PREFR, PREFI = 5, 7, 10

K_GEOPOTENTIAL, K_ZM_TRIGGER, K_ZM_PLUME, K_ZM_EVAP, K_MOMTRAN, K_AERO_SW, K_AERO_LW, K_UPDATE, K_CONVTRAN = range(9)
KERNEL_NAMES = ["geopotential_t", "zm_trigger", "zm_plume1", "zm_conv_evap", "momtran", "modal_aero_sw", "modal_aero_lw",
                "physics_update", "convtran", "zm_ascent2", "zm_plume2", "modal_aero_sw_diag"]


class ZmParams(C.Structure):
    _fields_ = (
        [(n, C.c_double) for n in (
            "cpair", "epsilo", "gravit", "latice", "latvap", "tmelt", "rair", "cpwv", "cpliq", "rh2o", "cpvir", "zvir",
            "c0_lnd", "c0_ocn", "ke", "ke_lnd", "momcu", "momcd", "tiedke_add", "capelmt", "dmpdz_param", "tau",
            "tiedke_lnd", "entrmn", "alfadet", "tentrm", "plclmin", "cin_threshd", "parcel_hscale")]
        + [(n, C.c_int32) for n in (
            "limcnv", "num_cin", "zm_org", "zmconv_microp", "no_deep_pbl", "lparcel_pbl", "cam3", "reserved")]
    )


def make_zm_params(physconst: dict, namelist: dict, limcnv: int) -> ZmParams:
    """What zm_convi receives (zm_conv.F90:115-120) + physconst + module defaults (:83-97,103)."""
    p = ZmParams()
    for k in ("cpair", "epsilo", "gravit", "latice", "latvap", "tmelt", "rair", "cpwv", "cpliq", "rh2o", "cpvir", "zvir"):
        setattr(p, k, physconst[k])
    for k in ("c0_lnd", "c0_ocn", "ke", "ke_lnd", "momcu", "momcd", "tiedke_add", "capelmt", "tau"):
        setattr(p, k, namelist[k])
    p.dmpdz_param = namelist["dmpdz"]
    p.tiedke_lnd = 1.0
    p.entrmn = 2.0e-4
    p.alfadet = 0.1
    p.tentrm = -namelist["dmpdz"]          # zm_convi resets tentrm = -dmpdz_param (:213)
    p.plclmin = 6.0e2
    p.cin_threshd = 0.33
    p.parcel_hscale = 0.5
    p.limcnv = limcnv
    p.num_cin = namelist["num_cin"]
    p.zm_org = namelist["org"]
    p.zmconv_microp = namelist["microp"]
    p.no_deep_pbl = namelist["no_deep_pbl"]
    p.lparcel_pbl = namelist["parcel_pbl"]
    p.cam3 = 0
    return p


class ZmConvrIO(C.Structure):
    _fields_ = (
        [(n, PD) for n in ("t", "qh", "pap", "dpp", "zm", "paph", "zi", "geos", "pblh", "tpert", "landfrac",
                           "qtnd", "heat", "cme", "zdu", "rprd", "dlf", "ql", "dif", "dnlf", "dnif", "eurt",
                           "mcon", "pflx", "prec", "jctop", "jcbot", "cape", "rliq", "rice",
                           "mu", "md", "du", "eu", "ed", "dp", "dsubcld")]
        + [(n, PI) for n in ("jt", "maxg", "ideep", "lengath", "status")]
    )


ZM_CONVR_IN_MID = ("t", "qh", "pap", "dpp", "zm")
ZM_CONVR_IN_INT = ("paph", "zi")
ZM_CONVR_IN_COL = ("geos", "pblh", "tpert", "landfrac")
ZM_CONVR_OUT_MID = ("qtnd", "heat", "cme", "zdu", "rprd", "dlf", "ql", "dif", "dnlf", "dnif", "eurt",
                    "mu", "md", "du", "eu", "ed", "dp")
ZM_CONVR_OUT_INT = ("mcon", "pflx")
ZM_CONVR_OUT_COL = ("prec", "jctop", "jcbot", "cape", "rliq", "rice", "dsubcld")
ZM_CONVR_OUT_ICOL = ("jt", "maxg", "ideep", "status")
ZM_CONVR_GATHERED = ("mu", "md", "du", "eu", "ed", "dp", "dsubcld", "jt", "maxg")


class ZmConvEvapIO(C.Structure):
    _fields_ = [(n, PD) for n in ("t", "pmid", "pdel", "q", "prdprec", "cldfrc", "landfrac", "prec",
                                  "tend_s", "tend_s_snwprd", "tend_s_snwevmlt", "tend_q", "ntprprd", "ntsnprd",
                                  "flxprec", "flxsnow", "snow")]


EVAP_IN_MID = ("t", "pmid", "pdel", "q", "prdprec", "cldfrc")
EVAP_OUT_MID = ("tend_s", "tend_s_snwprd", "tend_s_snwevmlt", "tend_q", "ntprprd", "ntsnprd")
EVAP_OUT_INT = ("flxprec", "flxsnow")


class MomtranIO(C.Structure):
    _fields_ = (
        [(n, PD) for n in ("winds", "mu", "md", "du", "eu", "ed", "dp", "dsubcld")]
        + [(n, PI) for n in ("jt", "mx", "ideep", "lengath")]
        + [(n, PD) for n in ("dqdt", "pguall", "pgdall", "icwu", "icwd", "seten")]
    )


MOMTRAN_OUT_3D = ("dqdt", "pguall", "pgdall", "icwu", "icwd")


class ConvtranIO(C.Structure):
    _fields_ = (
        [(n, PD) for n in ("q", "fracis", "mu", "md", "du", "eu", "ed", "dp", "dpdry")]
        + [(n, PI) for n in ("jt", "mx", "ideep", "lengath", "doconvtran", "isdry")]
        + [("dqdt", PD)]
    )


DERIVED_FIELDS_IN = ("pint", "lnpint", "t", "q", "qmin", "phis", "cpair", "rair", "zvir")
DERIVED_FIELDS_OUT = {"pdel": 1, "rpdel": 1, "pmid": 1, "lnpmid": 1, "zi": 2, "zm": 1, "s": 1}


class DerivedFieldsIO(C.Structure):
    _fields_ = [(n, PD) for n in DERIVED_FIELDS_IN + tuple(DERIVED_FIELDS_OUT)]


DRY_MASS_OUT = {"pdeldry": 1, "rpdeldry": 1, "pmiddry": 1, "lnpmiddry": 1, "pintdry": 2, "lnpintdry": 2, "psdry": 0}


class DryMassIO(C.Structure):
    _fields_ = [("pint", PD), ("pdel", PD), ("q", PD), ("isdry", PI), ("qmin", PD)] + [(n, PD) for n in DRY_MASS_OUT]


CONVECT_DIAG_FIELDS = ("cmfmc", "qc", "rliq", "cnt", "cnb", "cmfmc2", "qc2", "rprdsh", "rliq2", "cnt2", "cnb2", "precc", "snow",
                       "icwmr", "evapcsh", "rprddp", "pmid", "rprdtot", "pcnt", "pcnb")


class ConvectDiagIO(C.Structure):
    _fields_ = [(n, PD) for n in CONVECT_DIAG_FIELDS]


class ZmConvTend2IO(C.Structure):
    _fields_ = (
        [(n, PD) for n in ("q", "fracis", "pdeldry", "mu", "md", "du", "eu", "ed", "dp")]
        + [(n, PI) for n in ("jt", "maxg", "ideep", "lq", "isdry")]
        + [("ptend_q", PD)]
    )


# zm_conv_tend on resident data (colphys_zm_conv_tend_io): name -> (kind, levels); kind "d" double, "i" int32; levels 0 = [ncol],
# 1 = [ncol][pver], 2 = [ncol][pver+1], 3 = [nchunks]
ZM_TEND_IN = ("t", "q", "u", "v", "pmid", "pint", "pdel", "rpdel", "lnpint", "lnpmid", "zm", "zi", "phis", "ps", "rair", "zvir",
              "pblh", "tpert", "landfrac", "cld")
ZM_TEND_OUT = {
    "ptend_s": 1, "ptend_q": 1, "ptend_u": 1, "ptend_v": 1,
    "prec": 0, "snow": 0, "rliq": 0, "rice": 0, "jctop": 0, "jcbot": 0, "cape": 0,
    "ql": 1, "rprd": 1, "evapcdp": 1, "dlf": 1, "mconzm": 2, "flxprec": 2, "flxsnow": 2,
    "mu": 1, "md": 1, "du": 1, "eu": 1, "ed": 1, "dp": 1, "dsubcld": 0,
}
ZM_TEND_OUT_I = {"jt": 0, "maxg": 0, "ideep": 0, "lengath": 3}
ZM_TEND_DIAG = {"freqzm": 0, "pcont": 0, "pconb": 0, "mu_out": 1, "md_out": 1}
ZM_TEND_STATE1 = {"t1": 1, "q1": 1, "u1": 1, "v1": 1, "zi1": 2, "zm1": 1}
# ABI 4: zm_conv_tend's remaining intent(out) arguments and what it hands to outfld (4 = [2][ncol][pver], u and v)
ZM_TEND_HIST = {"cme": 1, "zdu": 1, "dif": 1, "eurt": 1, "pflx": 2, "zmdt_s": 1, "zmdq": 1, "evap_s": 1, "evap_s_snwprd": 1,
                "evap_s_snwevmlt": 1, "ntprprd": 1, "ntsnprd": 1, "seten": 1, "pguall": 4, "pgdall": 4, "icwu": 4, "icwd": 4}


class ZmConvTendIO(C.Structure):
    _fields_ = (
        [(n, PD) for n in ZM_TEND_IN]
        + [(n, PD) for n in ZM_TEND_OUT]
        + [(n, PI) for n in ZM_TEND_OUT_I]
        + [(n, PD) for n in ZM_TEND_DIAG]
        + [(n, PD) for n in ZM_TEND_STATE1]
        + [("status", PI)]
        + [(n, PD) for n in ZM_TEND_HIST]
    )


class AeroTables(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("nmodes", "nspec_max", "nswbands", "nlwbands", "top_lev",
                                  "idx_sw_diag", "idx_uv_diag", "idx_nir_diag")]
        + [(n, C.c_double) for n in ("rhoh2o", "rga", "rair", "xrmin", "xrmax")]
        + [("nspec", PI), ("spectype", PI)]
        + [(n, PD) for n in ("sigmag", "specdens", "hygro", "specrefindex_sw", "specrefindex_lw",
                             "refrtabsw", "refitabsw", "refrtablw", "refitablw",
                             "extpsw", "abspsw", "asmpsw", "absplw", "crefwsw", "crefwlw")]
    )


class ModalAeroIO(C.Structure):
    _fields_ = (
        [(n, PD) for n in ("pdeldry", "pmid", "t", "specmmr", "dgnumwet", "qaerwat")]
        + [("troplevchem", PI)]
        + [(n, PD) for n in ("tauxar", "wa", "ga", "fa")]
        + [("status", PI), ("vis_optics", PD)]
    )


# modal_aero_sw history diagnostics (include/colphys.h COLPHYS_SWD_*): field order of the packed outputs, with the
# reference's outfld names (modal_aer_opt.F90:1094-1154; the "dn" day+night set)
SWD_LEV_FIELDS = ("EXTINCTdn", "ABSORBdn", "EXTxASYMdn", "EXTINCTUVdn", "EXTINCTNIRdn")
SWD_COL_FIELDS = (
    "AODVISdn", "AODABSdn", "AODTOTdn", "AODVISstdn", "SSAVISdn", "AODxASYMdn", "AODUVdn", "AODNIRdn", "AODUVstdn",
    "AODNIRstdn", "BURDENSO4dn", "BURDENPOMdn", "BURDENSOAdn", "BURDENBCdn", "BURDENDUSTdn", "BURDENSEASALTdn",
    "AODABSBCdn", "AODSO4dn", "AODPOMdn", "AODSOAdn", "AODBCdn", "AODDUSTdn", "AODSSdn",
)
SWD_NLEV = len(SWD_LEV_FIELDS)
SWD_NCOL = len(SWD_COL_FIELDS)
SWD_NMODEF = 3          # per mode: BURDENdn<mm>, AODdn_<mode>, AODdnDUST<mm>


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.shape)
    return a.ctypes.data_as(PD)


def iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.shape)
    return a.ctypes.data_as(PI)


def complex_pairs(z: np.ndarray) -> np.ndarray:
    """complex array -> float64 [... , 2] (re, im)"""
    return np.ascontiguousarray(np.stack([z.real, z.imag], axis=-1), dtype=np.float64)


def build_aero_tables(props: dict, physconst: dict, keep: list) -> AeroTables:
    """Pack synth.make_aero_props()-style tables; `keep` receives the arrays that must outlive the struct."""
    import math
    tb = AeroTables()
    tb.nmodes = int(props["nmodes"]); tb.nspec_max = int(props["nspec_max"])
    tb.nswbands = props["crefwsw"].shape[0]; tb.nlwbands = props["crefwlw"].shape[0]
    tb.top_lev = int(props.get("top_lev", 1))
    tb.idx_sw_diag, tb.idx_uv_diag, tb.idx_nir_diag = 10, 11, 8
    tb.rhoh2o = physconst["rhoh2o"]; tb.rga = physconst["rga"]; tb.rair = physconst["rair"]
    tb.xrmin = math.log(0.01e-6); tb.xrmax = math.log(25.0e-6)       # modal_aer_opt.F90:140-143

    def f64(name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64); keep.append(a); setattr(tb, name, dptr(a))

    def i32(name, arr):
        a = np.ascontiguousarray(arr, dtype=np.int32); keep.append(a); setattr(tb, name, iptr(a))

    i32("nspec", props["nspec"]); i32("spectype", props["spectype"])
    for n in ("sigmag", "specdens", "hygro", "refrtabsw", "refitabsw", "refrtablw", "refitablw",
              "extpsw", "abspsw", "asmpsw", "absplw"):
        f64(n, props[n])
    f64("specrefindex_sw", complex_pairs(props["specrefindex_sw"]))
    f64("specrefindex_lw", complex_pairs(props["specrefindex_lw"]))
    f64("crefwsw", complex_pairs(props["crefwsw"]))
    f64("crefwlw", complex_pairs(props["crefwlw"]))
    return tb
