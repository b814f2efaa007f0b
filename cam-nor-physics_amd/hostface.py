"""Python mirror of the whole-rank HOST FACE (include/colphys.h, colphys_hostface_*): the chunk records of a rank in
pinned host memory, exposed as numpy views in the reference's own per-chunk layout -- Fortran (pcols, nlev), i.e. numpy
[nchunks][nlev][pcols] -- plus helpers that fill them from / read them into the column-contiguous arrays of
cam_nor_physics_amd.synth, which is what a restructured chunk loop's copy of state(c) into its record amounts to.

The Fortran face of the same C entry points is cam-nor-physics_amd/fortran/colphys_hostface_mod.F90.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import abi

HF_ZM, HF_AERO, HF_SWDIAG, HF_HIST = 1, 2, 4, 8


class HostfaceTimes(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("wall_ms", "host_cpu_ms", "h2d_ms", "relayout_ms", "physics_ms", "d2h_ms")] + \
               [("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64), ("nslices", C.c_int32), ("nstreams", C.c_int32)]


class HostFace:
    def __init__(self, eng, ncol_per_chunk, pcols, pver, what=HF_ZM, nslices=0):
        self.eng, self.lib = eng, eng.lib
        self.ncol = np.ascontiguousarray(ncol_per_chunk, dtype=np.int32)
        self.nchunks, self.pcols, self.pver, self.what = len(self.ncol), pcols, pver, what
        lib = self.lib
        lib.colphys_hostface_in_base.restype = C.c_void_p
        lib.colphys_hostface_out_base.restype = C.c_void_p
        lib.colphys_hostface_in_base.argtypes = [C.c_void_p]
        lib.colphys_hostface_out_base.argtypes = [C.c_void_p]
        lib.colphys_hostface_run.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double]
        lib.colphys_hostface_destroy.argtypes = [C.c_void_p]
        lib.colphys_hostface_last_times.argtypes = [C.c_void_p, C.POINTER(HostfaceTimes)]
        lib.colphys_hostface_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        lib.colphys_hostface_field.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        h = C.c_void_p()
        eng._check(lib.colphys_hostface_create(C.c_int(self.nchunks), C.c_int(pcols), C.c_int(pver), abi.iptr(self.ncol), C.c_int(what),
                                               C.c_int(nslices), C.byref(h)))
        self.h = h
        irec, orec, nf = C.c_int64(), C.c_int64(), C.c_int32()
        eng._check(lib.colphys_hostface_geometry(h, C.byref(irec), C.byref(orec), C.byref(nf)))
        self.in_rec, self.out_rec = irec.value, orec.value
        # the pinned slabs as byte arrays; per-field strided views [nchunks][nlev][pcols] on top of them
        self._in = np.ctypeslib.as_array((C.c_ubyte * (self.nchunks * self.in_rec)).from_address(lib.colphys_hostface_in_base(h)))
        self._out = np.ctypeslib.as_array((C.c_ubyte * (self.nchunks * self.out_rec)).from_address(lib.colphys_hostface_out_base(h)))
        self.fields = {}
        for i in range(nf.value):
            name = C.create_string_buffer(32)
            nlev, is_int, is_out, off = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
            eng._check(lib.colphys_hostface_field(h, i, name, C.byref(nlev), C.byref(is_int), C.byref(is_out), C.byref(off)))
            self.fields[name.value.decode()] = (nlev.value, bool(is_int.value), bool(is_out.value), off.value)

    def view(self, name):
        """numpy view of field `name` over all chunks: [nchunks][nlev][pcols] (Fortran (pcols, nlev) per chunk); lengath: [nchunks]"""
        nlev, is_int, is_out, off = self.fields[name]
        slab, rec = (self._out, self.out_rec) if is_out else (self._in, self.in_rec)
        dt = np.int32 if is_int else np.float64
        item = 4 if is_int else 8
        if nlev == 0:                                       # lengath: one int per chunk
            return np.lib.stride_tricks.as_strided(slab[off:].view(dt), shape=(self.nchunks,), strides=(rec,), writeable=True)
        return np.lib.stride_tricks.as_strided(slab[off:].view(dt), shape=(self.nchunks, nlev, self.pcols), strides=(rec, self.pcols * item, item),
                                               writeable=True)

    # ---- what a chunk loop's copies amount to, vectorised over the chunks -------------------------------------------------
    def put_columns(self, name, a, pad=np.inf):
        """a: column-contiguous [ncol_total][nlev] (or [ncol_total]) array of the rank's ACTIVE columns in chunk order; padding
        columns get `pad` (CAM leaves +inf there, physics_types.F90:2143-2178)"""
        v = self.view(name)
        nlev = v.shape[1]
        a = np.asarray(a).reshape(-1, nlev)
        full = (self.ncol == self.pcols)
        if full[:-1].all():                                 # ne30-shaped: only the last chunk is ragged
            nb = (self.nchunks - 1) * self.pcols if not full[-1] else self.nchunks * self.pcols
            v[:nb // self.pcols] = a[:nb].reshape(nb // self.pcols, self.pcols, nlev).transpose(0, 2, 1)
            if not full[-1]:
                n_last = int(self.ncol[-1])
                v[-1, :, :n_last] = a[nb:nb + n_last].T
                v[-1, :, n_last:] = pad if v.dtype == np.float64 else 0
            return
        c0 = 0
        for c, n in enumerate(self.ncol):
            v[c, :, :n] = a[c0:c0 + n].T
            v[c, :, n:] = pad if v.dtype == np.float64 else 0
            c0 += n

    def get_columns(self, name, gathered=False):
        """the ACTIVE columns of output `name` as [ncol_total][nlev] ([ncol_total] for per-column fields); gathered=True keeps the
        per-chunk gathered layout: rows c*pcols + j of a flat [nchunks*pcols][nlev] array, as the engine's *_dev outputs have it"""
        v = self.view(name)
        nlev = v.shape[1]
        if gathered:
            out = v.transpose(0, 2, 1).reshape(self.nchunks * self.pcols, nlev)
        else:
            out = np.concatenate([v[c, :, :n].T for c, n in enumerate(self.ncol)]) if not (self.ncol[:-1] == self.pcols).all() else \
                v.transpose(0, 2, 1).reshape(self.nchunks * self.pcols, nlev)[:(self.nchunks - 1) * self.pcols + int(self.ncol[-1])]
        return np.ascontiguousarray(out[:, 0] if nlev == 1 else out)

    def run(self, ztodt, dycore=abi.DYCORE_FV, qmin=1.0e-12):
        self.eng._check(self.lib.colphys_hostface_run(self.h, C.c_double(ztodt), C.c_int(dycore), C.c_double(qmin)))
        t = HostfaceTimes()
        self.eng._check(self.lib.colphys_hostface_last_times(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in HostfaceTimes._fields_}

    def close(self):
        if self.h:
            self._in = self._out = None
            self.lib.colphys_hostface_destroy(self.h)
            self.h = None

    # ---- filling from / comparing with the synthetic rank of cam_nor_physics_amd.synth ---------------------------------------
    def fill_from_synth(self, st, aer=None, props=None):
        if self.what & HF_ZM:
            for k in ("t", "q", "u", "v", "pmid", "pdel", "rpdel", "lnpmid", "zm", "cld", "pint", "lnpint", "zi", "phis", "pblh", "tpert", "landfrac"):
                self.put_columns(k, st[k])
            self.put_columns("ps", st["pint"][:, -1])
        else:                                               # the optics alone
            self.put_columns("t", st["t"]); self.put_columns("pmid", st["pmid"])
        if self.what & HF_AERO:
            self.put_columns("pdeldry", st["pdeldry"])
            for m in range(len(props["nspec"])):
                self.put_columns(f"dgnumwet_{m + 1}", aer["dgnumwet"][m])
                self.put_columns(f"qaerwat_{m + 1}", aer["qaerwat"][m])
                for l in range(int(props["nspec"][m])):
                    self.put_columns(f"specmmr_{m + 1}_{l + 1}", aer["specmmr"][m, l])
            if self.what & HF_SWDIAG:
                self.put_columns("troplevchem", np.asarray(aer["troplevchem"], dtype=np.int32), pad=0)


def chunk_counts(ncol, pcols):
    n = (ncol + pcols - 1) // pcols
    c = np.full(n, pcols, dtype=np.int32)
    c[-1] = ncol - (n - 1) * pcols
    return c


def measure_dropin(eng, st, aer, props, pcols=16, sw_diag=True, resident_ms=None, repeats=5, nslices=0):
    """bench.py's `dropin` leg: the same substep entered through the host face -- host records in the reference's chunk layout
    in, results out -- with the upload, the device re-layout, the kernels, the re-layout back and the download of the whole
    rank inside the timed region.  Returns the JSON object; raises if the face's results differ from nothing it can check
    cheaply here (the bit-for-bit comparison with the resident path is tests/test_hostface_gpu.py)."""
    ncol, pver = st["t"].shape
    what = HF_ZM | (HF_AERO if aer is not None else 0) | (HF_SWDIAG if (aer is not None and sw_diag) else 0)
    hf = HostFace(eng, chunk_counts(ncol, pcols), pcols, pver, what, nslices)
    try:
        t0 = time.perf_counter()
        hf.fill_from_synth(st, aer, props)
        pack_s = time.perf_counter() - t0
        from . import synth
        hf.run(synth.ZTODT)                                 # first run: work arenas are allocated
        runs = [hf.run(synth.ZTODT) for _ in range(repeats)]
        best = min(runs, key=lambda r: r["wall_ms"])
        gb = 1e-9
        h2d_gbs = best["h2d_bytes"] * gb / (best["h2d_ms"] * 1e-3) if best["h2d_ms"] > 0 else None
        d2h_gbs = best["d2h_bytes"] * gb / (best["d2h_ms"] * 1e-3) if best["d2h_ms"] > 0 else None
        # the transfers alone, one big copy each way, as the PCIe yardstick
        import torch
        hin = torch.empty(best["h2d_bytes"], dtype=torch.uint8).pin_memory()
        hout = torch.empty(best["d2h_bytes"], dtype=torch.uint8).pin_memory()
        din = torch.empty(best["h2d_bytes"], dtype=torch.uint8, device=eng.device)
        dout = torch.empty(best["d2h_bytes"], dtype=torch.uint8, device=eng.device)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        best_pcie = None
        for _ in range(3):
            t0 = time.perf_counter()
            with torch.cuda.stream(s1):
                din.copy_(hin, non_blocking=True)
            with torch.cuda.stream(s2):
                hout.copy_(dout, non_blocking=True)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) * 1e3
            best_pcie = el if best_pcie is None else min(best_pcie, el)
        lengath = hf.view("lengath")
        out = {
            "what": "host records (pcols,pver) per chunk in pinned memory -> H2D -> device re-layout -> zm_conv_tend"
                    + (" + modal_aero_sw" + (" (with diagnostics)" if sw_diag else "") + " + modal_aero_lw" if aer is not None else "")
                    + " -> device re-layout -> D2H, whole rank, pipelined over slices",
            "value": round(ncol / (best["wall_ms"] * 1e-3), 1), "unit": "columns/s", "wall_ms": round(best["wall_ms"], 3),
            "host_cpu_ms_to_enqueue": round(best["host_cpu_ms"], 3),
            "h2d_ms": round(best["h2d_ms"], 3), "relayout_ms": round(best["relayout_ms"], 3), "physics_ms": round(best["physics_ms"], 3),
            "d2h_ms": round(best["d2h_ms"], 3), "h2d_gb": round(best["h2d_bytes"] * gb, 4), "d2h_gb": round(best["d2h_bytes"] * gb, 4),
            "h2d_gb_per_s": round(h2d_gbs, 1) if h2d_gbs else None, "d2h_gb_per_s": round(d2h_gbs, 1) if d2h_gbs else None,
            "pcie_both_ways_alone_ms": round(best_pcie, 3), "wall_over_pcie": round(best["wall_ms"] / best_pcie, 3),
            "nslices": best["nslices"], "nstreams": best["nstreams"], "nchunks": hf.nchunks, "pcols": pcols,
            "host_pack_ms_numpy": round(pack_s * 1e3, 1),
            "deep_columns": int(lengath.sum()),
            "note": "phase times are summed over slices and overlap across the three streams; host_pack is THIS harness filling the records "
                    "with numpy (single thread) -- in the model it is the chunk loop's own copies of state(c), spread over its threads",
        }
        if resident_ms:
            out["resident_ms"] = round(resident_ms, 4)
        return out
    finally:
        hf.close()
