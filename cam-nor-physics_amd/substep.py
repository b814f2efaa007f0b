"""The whole-rank physics substep the benchmark times and the tests check -- ONE definition for both.

What tphysbc does chunk by chunk (physpkg.F90:1147-1161 -> :2820-2838 convect_deep_tend -> zm_conv_tend, and the two optics
calls radiation makes, modal_aer_opt.F90:485, 1216) becomes one dispatch over every column of the rank, resident in HBM:

    stream A (critical chain)   zm_conv_tend = zm_convr -> update -> geopotential_t -> zm_conv_evap -> update -> geopotential_t
                                               -> momtran -> update -> geopotential_t            (zm_conv_intr.F90:662-836)
    stream B (throughput work)  modal_aero_sw -> its history diagnostics -> modal_aero_lw

captured once into a HIP graph and replayed.  torch is plumbing only: device memory, the streams, the graph object.
"""
from __future__ import annotations

import numpy as np

from . import abi, synth

STATE_FIELDS = ("t", "q", "u", "v", "pint", "pmid", "pdel", "rpdel", "lnpint", "lnpmid", "pdeldry", "phis", "pblh", "tpert",
                "landfrac", "cld", "zi", "zm")


class ResidentSubstep:
    """st / aer / props: host dictionaries of cam_nor_physics_amd.synth for THIS rank's columns."""

    def __init__(self, eng, st, aer=None, props=None, *, pcols=16, sw_diag=True, two_streams=True, lw_kernel="auto",
                 dt=synth.ZTODT, rz_fields=False, lw_first=False):
        import torch
        self.torch = torch
        self.eng = eng
        self.ncol, self.pver = st["t"].shape
        self.pcols, self.dt = pcols, dt
        self.with_aero = aer is not None
        self.sw_diag = self.with_aero and sw_diag
        self.lw_first = lw_first                        # A/B: modal_aero_lw ahead of modal_aero_sw on the optics stream
        d = eng.dev
        self.S = {k: d(st[k]) for k in STATE_FIELDS}
        # rairv / zvirv (physics_types.F90:393-415) hold ONE value each unless WACCM-X varies the composition: the engine's
        # own constants serve (NULL fields, include/colphys.h); rz_fields=True uploads the two [ncol][pver] fields instead
        # (same bits, two more streams per geopotential_t)
        if rz_fields:
            self.S["rair"], self.S["zvir"] = d(st["rair"]), d(st["zvir"])
        self.S["ps"] = d(np.ascontiguousarray(st["pint"][:, -1]))
        self.zm_out = eng.alloc_zm_conv_tend_out(self.ncol, self.pver, pcols)
        self.side = None
        if self.with_aero:
            eng.aero_tables_set(props, synth.PHYSCONST)
            ncol, pver = self.ncol, self.pver
            self.A = {k: d(aer[k]) for k in ("specmmr", "dgnumwet", "qaerwat")}
            self.sw_out = {k: eng.empty(ncol, synth.NSWBANDS, pver + 1) for k in ("tauxar", "wa", "ga", "fa")}
            self.sw_out["status"] = eng.empty(ncol, dtype=torch.int32)
            self.lw_out = dict(tauxar=eng.empty(ncol, synth.NLWBANDS, pver), status=eng.empty(ncol, dtype=torch.int32))
            if self.sw_diag:
                nm = len(props["nspec"])
                self.A["troplevchem"] = d(aer["troplevchem"]).to(torch.int32)
                self.swd_vis = eng.empty(3, nm, ncol, pver)         # visible-band optics per mode, SW kernel -> its diagnostics
                self.swd_out = dict(lev=eng.empty(abi.SWD_NLEV, ncol, pver), col=eng.empty(abi.SWD_NCOL + abi.SWD_NMODEF * nm, ncol))
            # The optics share no data with the ZM chain: they go on a second stream and fill the SIMDs the latency-bound
            # chain leaves idle.
            if two_streams:
                self.side = torch.cuda.Stream(priority=0)
                self.step_start = torch.cuda.Event()
            eng.aero_kernel_select("auto", lw_kernel)
        self.graph = None
        self.clear_outputs()        # an element no kernel writes is NaN from the start, not whatever the allocator left

    # ---- the launches of one substep, on the current stream (+ the side stream) -------------------------------------
    def optics(self):
        eng, S, A = self.eng, self.S, self.A
        if self.lw_first:
            eng.modal_aero_lw(S["pdeldry"], A["specmmr"], A["dgnumwet"], A["qaerwat"], out=self.lw_out)
        eng.modal_aero_sw(S["pdeldry"], A["specmmr"], A["dgnumwet"], A["qaerwat"], out=self.sw_out, vis=self.swd_vis if self.sw_diag else None)
        if self.sw_diag:                                # what the reference's modal_aero_sw also computes, for outfld
            eng.modal_aero_sw_diag(S["pdeldry"], S["pmid"], S["t"], A["specmmr"], A["dgnumwet"], A["qaerwat"], A["troplevchem"],
                                   self.sw_out["tauxar"], out=self.swd_out, vis=self.swd_vis)
        if not self.lw_first:
            eng.modal_aero_lw(S["pdeldry"], A["specmmr"], A["dgnumwet"], A["qaerwat"], out=self.lw_out)

    def launch(self, overlap=True):
        """eager launches; overlap=False puts everything on the current stream (per-kernel timing passes)"""
        torch = self.torch
        side = self.side if overlap else None
        if side is not None:
            self.step_start.record()
        # the engine enqueues the trigger first; the optics go right behind it on the side stream so that the trigger's
        # long-running waves are placed before the optics' blocks
        self.eng.zm_conv_tend(self.S, self.dt, pcols=self.pcols, out=self.zm_out)
        if side is not None:
            side.wait_event(self.step_start)
            with torch.cuda.stream(side):
                self.optics()
            torch.cuda.current_stream().wait_stream(side)
        elif self.with_aero:
            self.optics()

    def capture(self):
        """captures the substep into a HIP graph (the ~20 launches are host-bound otherwise); returns self"""
        torch = self.torch
        self.launch()                                   # allocates the engine's workspaces before capture
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(priority=-1)            # the ZM chain is the critical path: high-priority queue
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            with torch.cuda.graph(self.graph, stream=cap):
                self.launch()
        torch.cuda.current_stream().wait_stream(cap)
        return self

    def release_graph(self):
        """a SECOND live graph of the same substep makes both replay slower: drop this one first"""
        self.graph = None
        self.torch.cuda.synchronize()

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.launch()

    # ---- results ----------------------------------------------------------------------------------------------------
    def outputs(self):
        o = {"zm." + k: v for k, v in self.zm_out.items()}
        if self.with_aero:
            o.update({"sw." + k: v for k, v in self.sw_out.items()})
            o.update({"lw." + k: v for k, v in self.lw_out.items()})
            if self.sw_diag:
                o.update({"swd." + k: v for k, v in self.swd_out.items()})
        return o

    def clear_outputs(self):
        """NaN / -1 in every output buffer, so that a replay that skips a kernel cannot pass a comparison on stale data"""
        for v in self.outputs().values():
            if v.dtype == self.torch.float64:
                v.fill_(float("nan"))
            else:
                v.fill_(-1)

    def fingerprint(self):
        """bitwise digest of every output (int64 views summed with wrap-around): equal digests <=> equal bits, up to collisions"""
        torch = self.torch
        torch.cuda.synchronize()
        parts = []
        for k, v in sorted(self.outputs().items()):
            w = v.contiguous().view(torch.int64) if v.dtype == torch.float64 else v.to(torch.int64)
            idx = torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.int64)
            parts.append((w.reshape(-1) * idx).sum())           # position-weighted: a permutation of values changes it
        return torch.stack(parts).cpu()
