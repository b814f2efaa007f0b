"""Host-side mirror of the reference's operator surface for the hot path, over the C ABI (include/colphys.h).

    reference (Fortran)                                      here
    -------------------------------------------------------  ---------------------------------
    zm_convi            physics/zm_conv.F90:115              ColumnPhysics(params, device)
    geopotential_t      physics/geopotential.F90:153         ColumnPhysics.geopotential_t
    zm_convr            physics/zm_conv.F90:231              ColumnPhysics.zm_convr
    zm_conv_evap        physics/zm_conv.F90:1712             ColumnPhysics.zm_conv_evap
    momtran             physics/zm_conv.F90:2315             ColumnPhysics.momtran
    modal_aero_sw / lw  physics/modal_aer_opt.F90:485,1216   ColumnPhysics.modal_aero_sw / _lw

Same argument names and meaning as the Fortran dummies; arrays are torch CUDA tensors in the
column-contiguous device layout ([ncol][pver], interfaces [ncol][pver+1]).  torch is plumbing only:
device memory and the current stream.  There is NO fallback: if libcolphys_hip.so is missing or no
GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi
from .build import LIB, build_hip

_lib = None


class ColphysError(RuntimeError):
    pass


def load_library(build_if_missing: bool = False) -> C.CDLL:
    """Loads the HIP extension; fails loudly when it is absent (no CPU path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime, and the process must hold ONE.  Loading this library before torch binds it
    # to /opt/rocm's runtime, and a later `import torch` then initialises a second one that owns the device
    # (colphys_init: "no HIP device") -- seen with build() followed by smoke() in one process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("COLPHYS_LIB", LIB)          # COLPHYS_LIB: A/B experiment builds only
    if not os.path.exists(path):
        if build_if_missing and path == LIB:
            build_hip()
        else:
            raise ColphysError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    lib.colphys_last_error.restype = C.c_char_p
    lib.colphys_init.argtypes = [C.c_int, C.POINTER(abi.ZmParams)]
    lib.colphys_profile_get.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "colphys_abi_version", "colphys_last_error", "colphys_device_count", "colphys_is_initialized", "colphys_zm_params_defaults",
    "colphys_init", "colphys_finalize",
    "colphys_geopotential_t_dev", "colphys_geopotential_t_chunks", "colphys_geopotential_dse_dev", "colphys_geopotential_dse_chunks",
    "colphys_zm_convr_dev", "colphys_zm_convr_chunks",
    "colphys_zm_conv_evap_dev", "colphys_zm_conv_evap_chunks",
    "colphys_momtran_dev", "colphys_momtran_chunks",
    "colphys_convtran_dev", "colphys_convtran_chunks", "colphys_physics_update_dev", "colphys_zm_conv_tend_dev",
    "colphys_zm_conv_tend_2_dev", "colphys_convect_diagnostics_dev", "colphys_derived_fields_dev", "colphys_dry_mass_fields_dev",
    "colphys_aero_tables_set",
    "colphys_modal_aero_sw_dev", "colphys_modal_aero_lw_dev",
    "colphys_modal_aero_sw_chunks", "colphys_modal_aero_lw_chunks",
    "colphys_modal_aero_sw_diag_dev", "colphys_modal_aero_sw_diag_chunks", "colphys_aero_grid_probe_dev", "colphys_aero_kernel_select",
    "colphys_profile_enable", "colphys_profile_reset", "colphys_profile_get", "colphys_math_probe_dev",
    "colphys_hostface_create", "colphys_hostface_destroy", "colphys_hostface_geometry", "colphys_hostface_field",
    "colphys_hostface_in_base", "colphys_hostface_out_base", "colphys_hostface_run", "colphys_hostface_last_times",
]


def _p(t, ctype=abi.PD):
    return C.cast(C.c_void_p(t.data_ptr()), ctype)


class ColumnPhysics:
    def __init__(self, params: abi.ZmParams, device: int = 0):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise ColphysError("no HIP device visible: the engine has no CPU path")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.params = params
        self._check(self.lib.colphys_init(device, C.byref(params)))
        self._keep = []
        self.tables = None

    # ---- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise ColphysError(f"colphys error {rc}: {self.lib.colphys_last_error().decode()}")

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def dev(self, a, dtype=None):
        """numpy / tensor -> contiguous CUDA tensor (fp64 unless int32 requested)"""
        t = self.torch
        if isinstance(a, np.ndarray):
            a = t.from_numpy(np.ascontiguousarray(a))
        if dtype is None:
            dtype = t.int32 if a.dtype in (t.int32, t.int64) else t.float64
        return a.to(device=self.device, dtype=dtype).contiguous()

    def _f64(self, x):
        assert x.is_cuda and x.dtype == self.torch.float64 and x.is_contiguous(), (x.dtype, x.device)
        return x

    def empty(self, *shape, dtype=None):
        return self.torch.empty(*shape, dtype=dtype or self.torch.float64, device=self.device)

    def close(self):
        self.lib.colphys_finalize()

    # ---- geopotential_t (geopotential.F90:153) ----------------------------------------------------
    def geopotential_t(self, piln, pmln, pint, pmid, pdel, rpdel, t, q, rair, gravit, zvir, zi=None, zm=None,
                       dycore=abi.DYCORE_FV, nspecies=1):
        """q: [ncol][pver] (water vapour) or [nq][ncol][pver]."""
        n, pver = t.shape
        nq = 1 if q.dim() == 2 else q.shape[0]
        zi = self.empty(n, pver + 1) if zi is None else zi
        zm = self.empty(n, pver) if zm is None else zm

        def opt(x):
            return _p(self._f64(x)) if x is not None else None
        self._check(self.lib.colphys_geopotential_t_dev(
            C.c_int(n), C.c_int(pver), C.c_int(dycore), C.c_int(nq), C.c_int(nspecies), opt(piln), opt(pmln), opt(pint),
            opt(pmid), opt(pdel), opt(rpdel), opt(t), opt(q), opt(rair), C.c_double(gravit), opt(zvir), _p(zi), _p(zm),
            self._stream()))
        return zi, zm

    # ---- geopotential_dse (geopotential.F90:29) ------------------------------------------------------
    def geopotential_dse(self, piln, pmln, pint, pmid, pdel, rpdel, dse, q, phis, rair, gravit, cpair, zvir,
                         dycore=abi.DYCORE_FV, nspecies=1):
        n, pver = dse.shape
        nq = 1 if q.dim() == 2 else q.shape[0]
        t, zi, zm = self.empty(n, pver), self.empty(n, pver + 1), self.empty(n, pver)

        def opt(x):
            return _p(self._f64(x)) if x is not None else None
        self._check(self.lib.colphys_geopotential_dse_dev(
            C.c_int(n), C.c_int(pver), C.c_int(dycore), C.c_int(nq), C.c_int(nspecies), opt(piln), opt(pmln), opt(pint),
            opt(pmid), opt(pdel), opt(rpdel), opt(dse), opt(q), opt(phis), opt(rair), C.c_double(gravit), opt(cpair), opt(zvir),
            _p(t), _p(zi), _p(zm), self._stream()))
        return t, zi, zm

    # ---- zm_convr (zm_conv.F90:231) ----------------------------------------------------------------
    def alloc_zm_convr_out(self, n, pver, pcols):
        t = self.torch
        o = {}
        for k in abi.ZM_CONVR_OUT_MID:
            o[k] = self.empty(n, pver)
        for k in abi.ZM_CONVR_OUT_INT:
            o[k] = self.empty(n, pver + 1)
        for k in abi.ZM_CONVR_OUT_COL:
            o[k] = self.empty(n)
        for k in abi.ZM_CONVR_OUT_ICOL:
            o[k] = self.empty(n, dtype=t.int32)
        o["lengath"] = self.empty((n + pcols - 1) // pcols, dtype=t.int32)
        return o

    def zm_convr(self, t, qh, pblh, zm, geos, zi, pap, paph, dpp, delt, tpert, landfrac, pcols=16, out=None):
        n, pver = t.shape
        o = self.alloc_zm_convr_out(n, pver, pcols) if out is None else out
        io = abi.ZmConvrIO()
        ins = dict(t=t, qh=qh, pap=pap, dpp=dpp, zm=zm, paph=paph, zi=zi, geos=geos, pblh=pblh, tpert=tpert, landfrac=landfrac)
        for k, v in ins.items():
            setattr(io, k, _p(self._f64(v)))
        for k, v in o.items():
            setattr(io, k, _p(v, abi.PI) if v.dtype == self.torch.int32 else _p(v))
        self._check(self.lib.colphys_zm_convr_dev(C.c_int(n), C.c_int(pver), C.c_int(pcols), C.c_double(delt), C.byref(io),
                                                  self._stream()))
        return o

    # ---- zm_conv_evap (zm_conv.F90:1712) -----------------------------------------------------------
    def zm_conv_evap(self, t, pmid, pdel, q, landfrac, prdprec, cldfrc, deltat, prec, out=None):
        """prec is updated in place (m/s in, m/s out) exactly like the reference's intent(inout) dummy."""
        n, pver = t.shape
        if out is None:
            out = {k: self.empty(n, pver) for k in abi.EVAP_OUT_MID}
            out.update({k: self.empty(n, pver + 1) for k in abi.EVAP_OUT_INT})
            out["snow"] = self.empty(n)
        io = abi.ZmConvEvapIO()
        ins = dict(t=t, pmid=pmid, pdel=pdel, q=q, prdprec=prdprec, cldfrc=cldfrc, landfrac=landfrac, prec=prec)
        for k, v in {**ins, **out}.items():
            setattr(io, k, _p(self._f64(v)))
        self._check(self.lib.colphys_zm_conv_evap_dev(C.c_int(n), C.c_int(pver), C.c_double(deltat), C.byref(io), self._stream()))
        out["prec"] = prec
        return out

    # ---- momtran (zm_conv.F90:2315) ----------------------------------------------------------------
    def momtran(self, winds, mu, md, du, eu, ed, dp, dsubcld, jt, mx, ideep, lengath, dt, pcols=16, out=None):
        """winds: [2][ncol][pver] (u, v)."""
        _, n, pver = winds.shape
        if out is None:
            out = {k: self.empty(2, n, pver) for k in abi.MOMTRAN_OUT_3D}
            out["seten"] = self.empty(n, pver)
        io = abi.MomtranIO()
        for k, v in dict(winds=winds, mu=mu, md=md, du=du, eu=eu, ed=ed, dp=dp, dsubcld=dsubcld).items():
            setattr(io, k, _p(self._f64(v)))
        for k, v in dict(jt=jt, mx=mx, ideep=ideep, lengath=lengath).items():
            assert v.dtype == self.torch.int32 and v.is_cuda
            setattr(io, k, _p(v, abi.PI))
        for k, v in out.items():
            setattr(io, k, _p(self._f64(v)))
        self._check(self.lib.colphys_momtran_dev(C.c_int(n), C.c_int(pver), C.c_int(pcols), C.c_double(dt), C.byref(io),
                                                 self._stream()))
        return out

    # ---- zm_conv_tend on resident data (zm_conv_intr.F90:441-905) -------------------------------------
    def alloc_zm_conv_tend_out(self, n, pver, pcols, hist=False):
        """hist: also the ABI-4 history / argument outputs (abi.ZM_TEND_HIST), which otherwise stay in the engine's work arrays"""
        t = self.torch
        shape = {0: (n,), 1: (n, pver), 2: (n, pver + 1), 3: ((n + pcols - 1) // pcols,), 4: (2, n, pver)}
        fields = {**abi.ZM_TEND_OUT, **abi.ZM_TEND_DIAG, **abi.ZM_TEND_STATE1, **(abi.ZM_TEND_HIST if hist else {})}
        o = {k: self.empty(*shape[lv]) for k, lv in fields.items()}
        o.update({k: self.empty(*shape[lv], dtype=t.int32) for k, lv in abi.ZM_TEND_OUT_I.items()})
        o["status"] = self.empty(n, dtype=t.int32)
        return o

    def zm_conv_tend(self, state: dict, ztodt, pcols=16, dycore=abi.DYCORE_FV, qmin=1.0e-12, out=None, hist=False):
        """state: device tensors named as abi.ZM_TEND_IN (rair / zvir may be absent: uniform); returns ptend_all, the pbuf fields, diagnostics and state1
        (hist: and the history / argument outputs of abi.ZM_TEND_HIST)."""
        n, pver = state["t"].shape
        o = self.alloc_zm_conv_tend_out(n, pver, pcols, hist) if out is None else out
        io = abi.ZmConvTendIO()
        for k in abi.ZM_TEND_IN:
            if k in ("rair", "zvir") and state.get(k) is None:
                continue                        # NULL: the engine's own constants (no WACCM-X; include/colphys.h)
            setattr(io, k, _p(self._f64(state[k])))
        for k, v in o.items():
            setattr(io, k, _p(v, abi.PI) if v.dtype == self.torch.int32 else _p(v))
        self._check(self.lib.colphys_zm_conv_tend_dev(C.c_int(n), C.c_int(pver), C.c_int(pcols), C.c_double(ztodt),
                                                      C.c_int(dycore), C.c_double(qmin), C.byref(io), self._stream()))
        return o

    # ---- convtran (zm_conv.F90:1976) ---------------------------------------------------------------
    def convtran(self, q, fracis, mu, md, du, eu, ed, dp, dpdry, jt, mx, ideep, lengath, doconvtran, isdry, dt, pcols=16,
                 dqdt=None):
        """q, fracis, dqdt: [ncnst][ncol][pver]; doconvtran / isdry: host sequences of ncnst flags."""
        ncnst, n, pver = q.shape
        if dqdt is None:
            dqdt = self.torch.zeros_like(q)
        io = abi.ConvtranIO()
        for k, v in dict(q=q, fracis=fracis, mu=mu, md=md, du=du, eu=eu, ed=ed, dp=dp, dpdry=dpdry, dqdt=dqdt).items():
            setattr(io, k, _p(self._f64(v)))
        for k, v in dict(jt=jt, mx=mx, ideep=ideep, lengath=lengath).items():
            assert v.dtype == self.torch.int32 and v.is_cuda
            setattr(io, k, _p(v, abi.PI))
        do = np.ascontiguousarray(doconvtran, dtype=np.int32); dry = np.ascontiguousarray(isdry, dtype=np.int32)
        io.doconvtran = abi.iptr(do); io.isdry = abi.iptr(dry)
        self._check(self.lib.colphys_convtran_dev(C.c_int(n), C.c_int(pver), C.c_int(pcols), C.c_int(ncnst), C.c_double(dt),
                                                  C.byref(io), self._stream()))
        return dqdt

    # ---- zm_conv_tend_2 / convtran1 on resident data (zm_conv_intr.F90:955-1028, 860-884) ---------------
    def zm_conv_tend_2(self, q, fracis, pdeldry, mu, md, du, eu, ed, dp, jt, maxg, ideep, lq, isdry, ztodt, pcols=16, ptend_q=None):
        """q, fracis, ptend_q: [ncnst][ncol][pver]; lq / isdry: host sequences of ncnst flags; pdeldry may be None when no
        selected constituent is 'dry'."""
        ncnst, n, pver = q.shape
        if ptend_q is None:
            ptend_q = self.torch.zeros_like(q)
        io = abi.ZmConvTend2IO()
        for k, v in dict(q=q, fracis=fracis, mu=mu, md=md, du=du, eu=eu, ed=ed, dp=dp, ptend_q=ptend_q).items():
            setattr(io, k, _p(self._f64(v)))
        io.pdeldry = _p(self._f64(pdeldry)) if pdeldry is not None else None
        for k, v in dict(jt=jt, maxg=maxg, ideep=ideep).items():
            assert v.dtype == self.torch.int32 and v.is_cuda
            setattr(io, k, _p(v, abi.PI))
        lqa = np.ascontiguousarray(lq, dtype=np.int32); dry = np.ascontiguousarray(isdry, dtype=np.int32)
        io.lq = abi.iptr(lqa); io.isdry = abi.iptr(dry)
        self._check(self.lib.colphys_zm_conv_tend_2_dev(C.c_int(n), C.c_int(pver), C.c_int(pcols), C.c_int(ncnst),
                                                        C.c_double(ztodt), C.byref(io), self._stream()))
        return ptend_q

    # ---- d_p_coupling derived fields (fv/dp_coupling.F90:550-629) ---------------------------------------
    def derived_fields(self, pint, lnpint, t, q, qmin, phis, cpair, rair, zvir, gravit, dycore=abi.DYCORE_FV, nspecies=1):
        """q: [ncnst][ncol][pver], fixed in place; qmin: host sequence of ncnst; returns the seven derived fields"""
        ncnst, n, pver = q.shape
        shape = {1: (n, pver), 2: (n, pver + 1)}
        out = {k: self.empty(*shape[lv]) for k, lv in abi.DERIVED_FIELDS_OUT.items()}
        io = abi.DerivedFieldsIO()
        qm = np.ascontiguousarray(qmin, dtype=np.float64)
        for k, v in dict(pint=pint, lnpint=lnpint, t=t, q=q, phis=phis, cpair=cpair, rair=rair, zvir=zvir, **out).items():
            setattr(io, k, _p(self._f64(v)))
        io.qmin = abi.dptr(qm)
        self._check(self.lib.colphys_derived_fields_dev(C.c_int(n), C.c_int(pver), C.c_int(ncnst), C.c_int(dycore), C.c_int(nspecies),
                                                        C.c_double(gravit), C.byref(io), self._stream()))
        return out

    # ---- set_state_pdry + set_wet_to_dry + qneg3 (fv/dp_coupling.F90:630-651, physics_types.F90:1925-1984) -----------
    def dry_mass_fields(self, pint, pdel, q, isdry, qmin):
        """q: [ncnst][ncol][pver], converted / clipped in place; isdry, qmin: host sequences of ncnst; returns the dry-pressure fields"""
        ncnst, n, pver = q.shape
        shape = {0: (n,), 1: (n, pver), 2: (n, pver + 1)}
        out = {k: self.empty(*shape[lv]) for k, lv in abi.DRY_MASS_OUT.items()}
        io = abi.DryMassIO()
        for k, v in dict(pint=pint, pdel=pdel, q=q, **out).items():
            setattr(io, k, _p(self._f64(v)))
        dry = np.ascontiguousarray(isdry, dtype=np.int32); qm = np.ascontiguousarray(qmin, dtype=np.float64)
        io.isdry = abi.iptr(dry); io.qmin = abi.dptr(qm)
        self._check(self.lib.colphys_dry_mass_fields_dev(C.c_int(n), C.c_int(pver), C.c_int(ncnst), C.byref(io), self._stream()))
        return out

    # ---- convect_diagnostics_calc (convect_diagnostics.F90:115-249) --------------------------------------
    def convect_diagnostics(self, fields: dict, shallow_is_clubb: bool):
        """fields: device tensors named as abi.CONVECT_DIAG_FIELDS (missing / None = NULL); updates in place, fills rprdtot, pcnt, pcnb"""
        n, pver = fields["qc"].shape
        io = abi.ConvectDiagIO()
        for k in abi.CONVECT_DIAG_FIELDS:
            v = fields.get(k)
            setattr(io, k, _p(self._f64(v)) if v is not None else None)
        self._check(self.lib.colphys_convect_diagnostics_dev(C.c_int(n), C.c_int(pver), C.c_int(1 if shallow_is_clubb else 0),
                                                             C.byref(io), self._stream()))
        return fields

    # ---- physics_update, tendency application (physics_types.F90:292-340,425-432) ---------------------
    def physics_update(self, dt, cpair, qmin=1.0e-12, t=None, q=None, u=None, v=None):
        """each of t, q, u, v is None or a triple (in, tend, out) of [ncol][pver] tensors"""
        first = next(x for x in (t, q, u, v) if x is not None)
        n, pver = first[0].shape
        ptrs = []
        for tri in (t, q, u, v):
            ptrs += [None, None, None] if tri is None else [_p(self._f64(x)) for x in tri]
        self._check(self.lib.colphys_physics_update_dev(C.c_int(n), C.c_int(pver), C.c_double(dt), C.c_double(cpair),
                                                        C.c_double(qmin), *ptrs, self._stream()))

    # ---- diagnostic: the device log / log10 / svp_water the kernels use ------------------------------
    PROBE = {"log": 0, "log10": 1, "svp_water": 2, "exp": 3}

    def math_probe(self, kind: str, x):
        x = self._f64(x)
        y = self.torch.empty_like(x)
        self._check(self.lib.colphys_math_probe_dev(C.c_int(self.PROBE[kind]), C.c_int64(x.numel()), _p(x), _p(y),
                                                    self._stream()))
        return y

    # ---- modal aerosol optics (modal_aer_opt.F90:485,1216) ------------------------------------------
    def aero_tables_set(self, props: dict, physconst: dict):
        self._keep = []
        self.tables = abi.build_aero_tables(props, physconst, self._keep)
        self._check(self.lib.colphys_aero_tables_set(C.byref(self.tables)))

    def _aero_io(self, pdeldry, specmmr, dgnumwet, qaerwat, pmid=None, t=None):
        io = abi.ModalAeroIO()
        io.pdeldry = _p(self._f64(pdeldry)); io.specmmr = _p(self._f64(specmmr))
        io.dgnumwet = _p(self._f64(dgnumwet)); io.qaerwat = _p(self._f64(qaerwat))
        io.pmid = _p(pmid) if pmid is not None else None
        io.t = _p(t) if t is not None else None
        return io

    def modal_aero_sw(self, pdeldry, specmmr, dgnumwet, qaerwat, out=None, vis=None):
        """vis: optional [3][nmodes][n][pver] buffer for the visible-band optics per mode, which modal_aero_sw_diag then reads
        instead of evaluating the tables again (colphys_modal_aero_io.vis_optics)"""
        n, pver = pdeldry.shape
        nsw = self.tables.nswbands
        if out is None:
            out = {k: self.empty(n, nsw, pver + 1) for k in ("tauxar", "wa", "ga", "fa")}
            out["status"] = self.empty(n, dtype=self.torch.int32)
        io = self._aero_io(pdeldry, specmmr, dgnumwet, qaerwat)
        io.vis_optics = _p(self._f64(vis)) if vis is not None else None
        for k in ("tauxar", "wa", "ga", "fa"):
            setattr(io, k, _p(out[k]))
        io.status = _p(out["status"], abi.PI)
        self._check(self.lib.colphys_modal_aero_sw_dev(C.c_int(n), C.c_int(pver), C.byref(io), self._stream()))
        return out

    def modal_aero_sw_diag(self, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplevchem, tauxar, out=None, vis=None):
        """modal_aero_sw's history diagnostics (modal_aer_opt.F90:896-993, 1046-1154), after modal_aero_sw has filled
        `tauxar`.  Returns lev [SWD_NLEV][n][pver] and col [SWD_NCOL + 3*nmodes][n] in the field order of
        abi.SWD_LEV_FIELDS / abi.SWD_COL_FIELDS (+ burden, aodmode, dustaodmode per mode)."""
        n, pver = pdeldry.shape
        if out is None:
            out = dict(lev=self.empty(abi.SWD_NLEV, n, pver), col=self.empty(abi.SWD_NCOL + abi.SWD_NMODEF * self.tables.nmodes, n))
        io = self._aero_io(pdeldry, specmmr, dgnumwet, qaerwat, self._f64(pmid), self._f64(t))
        io.troplevchem = _p(troplevchem, abi.PI)
        io.tauxar = _p(self._f64(tauxar))
        io.vis_optics = _p(self._f64(vis)) if vis is not None else None
        self._check(self.lib.colphys_modal_aero_sw_diag_dev(C.c_int(n), C.c_int(pver), C.byref(io), _p(out["lev"]), _p(out["col"]),
                                                            self._stream()))
        return out

    def modal_aero_lw(self, pdeldry, specmmr, dgnumwet, qaerwat, out=None):
        n, pver = pdeldry.shape
        nlw = self.tables.nlwbands
        if out is None:
            out = dict(tauxar=self.empty(n, nlw, pver), status=self.empty(n, dtype=self.torch.int32))
        io = self._aero_io(pdeldry, specmmr, dgnumwet, qaerwat)
        io.tauxar = _p(out["tauxar"]); io.status = _p(out["status"], abi.PI)
        self._check(self.lib.colphys_modal_aero_lw_dev(C.c_int(n), C.c_int(pver), C.byref(io), self._stream()))
        return out

    AERO_KERNEL = {"auto": 0, "bands-outer": 1, "grouped": 2}

    def aero_kernel_select(self, sw: str = "auto", lw: str = "auto"):
        """kernel family of modal_aero_sw / modal_aero_lw (include/colphys.h colphys_aero_kernel_select)"""
        self._check(self.lib.colphys_aero_kernel_select(C.c_int(self.AERO_KERNEL[sw]), C.c_int(self.AERO_KERNEL[lw])))

    def aero_grid_probe(self, sw: bool, mode: int, band: int, which: int, use_lut: bool, x):
        """table_interp's interval search of one grid (which: 0 real, 1 imaginary index) on device values x ->
        (ix1, ix2, t, lut_available); diagnostic (include/colphys.h colphys_aero_grid_probe_dev)"""
        x = self._f64(x)
        n = x.numel()
        ix1 = self.empty(n, dtype=self.torch.int32); ix2 = self.empty(n, dtype=self.torch.int32); t = self.empty(n)
        avail = C.c_int32(0)
        self._check(self.lib.colphys_aero_grid_probe_dev(C.c_int(int(sw)), C.c_int(mode), C.c_int(band), C.c_int(which),
                                                         C.c_int(int(use_lut)), C.c_int64(n), _p(x), _p(ix1, abi.PI), _p(ix2, abi.PI),
                                                         _p(t), C.byref(avail), self._stream()))
        return ix1, ix2, t, bool(avail.value)

    # ---- measurement ---------------------------------------------------------------------------------
    def profile_enable(self, on: bool):
        self.lib.colphys_profile_enable(int(on))

    def profile_reset(self):
        self.lib.colphys_profile_reset()

    def profile(self) -> dict:
        res = {}
        for i, name in enumerate(abi.KERNEL_NAMES):
            ms, n = C.c_double(), C.c_int64()
            self.lib.colphys_profile_get(i, C.byref(ms), C.byref(n))
            res[name] = (ms.value, n.value)
        return res
