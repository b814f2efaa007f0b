#!/usr/bin/env python3
"""Benchmark of the hot path: columns/sec for one "ZM + geopotential + aerosol-optics" physics substep on
synthetic ne30 x L32-shaped input (BASELINE.json: 48 602 columns x 32 levels), fp64.

One step = one pass of the substep over this rank's columns, inputs already resident in HBM:

    zm_convr -> physics_update -> geopotential_t -> zm_conv_evap -> physics_update -> geopotential_t
             -> momtran -> physics_update -> geopotential_t -> modal_aero_sw -> modal_aero_lw

(the call order of zm_conv_tend, zm_conv_intr.F90:662-836, followed by the two optics calls radiation makes,
modal_aer_opt.F90:485,1216).  Algorithmic bytes of the substep: 52 128 B/column at L32.

    python bench.py [--gpus N --steps K --warmup W]                 # N = 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU, static column split, NO collective on the data path; the only
communication is the barrier / max-reduce that brackets the timed region.  Default is weak scaling (every rank
owns a full 48 602-column ne30 slab, value = all ranks' columns / time); --strong splits ONE 48 602-column slab
across the ranks instead (the literal reading of BASELINE.json's metric).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on its launch stream) and
`cpu_baseline` (the reference Fortran, or the C restatement if oracle/_ref is absent, on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# algorithmic bytes per column (fp64), SURVEY.md 8d, as functions of pver = L
BYTES = {
    "geopotential_t": lambda L: 4 * L * 8 + (2 * L + 1) * 8,
    "zm_convr": lambda L: (5 * L + 2 * (L + 1) + 4) * 8 + (17 * L + 2 * (L + 1) + 7) * 8 + 12,
    "zm_conv_evap": lambda L: (6 * L + 2) * 8 + (6 * L + 2 * (L + 1) + 2) * 8,
    "momtran": lambda L: (8 * L + 1) * 8 + 12 + 11 * L * 8,
    "modal_aero_sw": lambda L: 26 * L * 8 + 4 * (L + 1) * 14 * 8,
    "modal_aero_lw": lambda L: 26 * L * 8 + 16 * L * 8,
    # history diagnostics of modal_aero_sw: the optics inputs once more + pmid, t, the UV / near-IR / all-band tauxar read
    # back, 5 level fields and 35 column fields out
    "modal_aero_sw_diag": lambda L: (26 + 2) * L * 8 + 14 * (L + 1) * 8 + 5 * L * 8 + 35 * 8,
}


def substep_bytes(L, sw_diag=True):
    return (BYTES["zm_convr"](L) + BYTES["zm_conv_evap"](L) + BYTES["momtran"](L) + 3 * BYTES["geopotential_t"](L)
            + BYTES["modal_aero_sw"](L) + BYTES["modal_aero_lw"](L) + (BYTES["modal_aero_sw_diag"](L) if sw_diag else 0))


PMC_KERNEL = {"geopotential_t": "geopotential_kernel", "zm_trigger": "zm_trigger_kernel", "zm_plume1": "zm_plume1_kernel",
              "zm_ascent2": "zm_ascent2_kernel", "zm_plume2": "zm_plume2_kernel",
              "zm_conv_evap": "zm_conv_evap_kernel", "momtran": "momtran_wave_kernel", "modal_aero_sw": "modal_aero_reg_kernel<true,",
              "modal_aero_lw": "modal_aero_reg_kernel<false,", "modal_aero_sw_diag": "modal_aero_sw_diag_vis_kernel"}


def pmc_of(kernel):
    """PMC-derived figures of `kernel` from the committed passes of this same command (scripts/pmc_passes.sh ->
    profiles/*_pmc_summary.csv), or None:
      traffic  HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE reads half the bytes of a coalesced
               stream on gfx950 (MI355X_MICROARCH.md, HBM); calibrated here on physics_update_kernel, whose byte count
               is known exactly;
      valu     SIMD VALU utilisation = SQ_ACTIVE_INST_VALU * 4 cycles / (kernel cycles * 1 024 SIMDs): what bounds this
               path (fp64 vector ALU), reported beside the HBM fraction the contract asks for."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")))
    if not files:
        return None
    for r in csv.DictReader(open(files[-1])):
        if r["kernel"].startswith(PMC_KERNEL.get(kernel, "?")) and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
            out = {"traffic": round((2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024 / 1e9, 4),
                   "source": os.path.basename(files[-1])}
            if r.get("SQ_ACTIVE_INST_VALU") and r.get("GRBM_GUI_ACTIVE"):
                out["valu"] = round(float(r["SQ_ACTIVE_INST_VALU"]) * 4.0 / (float(r["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0), 3)
            return out
    return None


def valu_issue_of_substep(ms_per_step, with_aero):
    """Substep-level fp64 VALU issue figure from the committed PMC passes: sum over the substep's launches of the wave-level
    VALU instructions (SQ_INSTS_VALU) x 4 cycles (one fp64 instruction per SIMD per 4 cycles, scripts/ubench) over
    1 024 SIMDs at the 2.4 GHz peak clock, as a fraction of the measured step time -- what bounds this path beside the HBM
    fraction the contract asks for."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")))
    if not files:
        return None
    mult = {"geopotential_kernel": 3, "physics_update_kernel": 3}
    total = 0.0
    for r in csv.DictReader(open(files[-1])):
        k = r["kernel"]
        if not r.get("SQ_INSTS_VALU") or (not with_aero and k.startswith("modal_aero")):
            continue
        total += float(r["SQ_INSTS_VALU"]) * mult.get(k, 1)
    ms_bound = total * 4.0 / (1024 * 2.4e9) * 1e3
    return {"wave_valu_instructions": round(total), "ms_at_full_issue": round(ms_bound, 4), "frac_of_step": round(ms_bound / ms_per_step, 4),
            "source": os.path.basename(files[-1])}


def baseline_metric():
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "columns/sec, ZM+aer-opt physics substep, 48602 cols x L32 at 1/2/4/8 MI355X"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ncol", type=int, default=48602)
    ap.add_argument("--pver", type=int, default=32)
    ap.add_argument("--strong", action="store_true", help="split ONE --ncol slab over the ranks")
    ap.add_argument("--no-aero", action="store_true", help="configs[1]: ZM only (zm_convr + zm_conv_evap + momtran)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying a captured HIP graph")
    ap.add_argument("--no-sw-diag", action="store_true",
                    help="leave out modal_aero_sw's history diagnostics (modal_aer_opt.F90:896-993, 1046-1209), which the "
                         "reference computes inside modal_aero_sw on every call: radiative outputs only")
    ap.add_argument("--lw-kernel", choices=["auto", "bands-outer", "grouped"], default="auto",
                    help="kernel family of modal_aero_lw (colphys_aero_kernel_select).  auto = bands-outer: with the SW diagnostics "
                         "in the substep the optics stream ends last and the kernel that is faster alone wins (2.425 against 2.453 ms); "
                         "with --no-sw-diag `grouped` measures 0.6 %% faster")
    ap.add_argument("--serial", action="store_true", help="all kernels on one stream (no ZM || optics overlap)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from cam_nor_physics_amd import synth
    from cam_nor_physics_amd.dist import shard_range

    pver = args.pver
    with_aero = not args.no_aero
    sw_diag = with_aero and not args.no_sw_diag
    if args.strong:
        lo, hi = shard_range(args.ncol, world, rank)
        st_all = synth.make_state(args.ncol, pver, synth.DEFAULT_SEED)
        st = {k: (v[lo:hi] if hasattr(v, "shape") and v.shape[:1] == (args.ncol,) else v) for k, v in st_all.items()}
        st["ncol"] = hi - lo
    else:
        st = synth.make_state(args.ncol, pver, synth.DEFAULT_SEED + rank)
    ncol = st["ncol"]
    props = synth.make_aero_props(synth.DEFAULT_SEED)
    aer = synth.make_aero_state(ncol, pver, props, synth.DEFAULT_SEED + rank)

    # ---- CPU baseline: rank 0 at N = 1 only, BEFORE this process touches the GPU (workers are forked) ----
    cpu = cpu_port = None
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from cpu_substep import run_cpu_baseline
        cpu = run_cpu_baseline(st, aer, props, pver, with_aero=with_aero)
        # the C restatement on the same cores as well: oracle/_ref is the reference's own Fortran, but built against
        # stand-in modules for the parent model it does not contain, so both CPU figures are reported
        cpu_port = run_cpu_baseline(st, aer, props, pver, with_aero=with_aero, kind="port") if cpu["kind"] == "reference" else None

    import torch
    import torch.distributed as dist
    from cam_nor_physics_amd import abi
    from cam_nor_physics_amd.engine import ColumnPhysics

    # Rehearsal switches (tests/test_bench_multirank_gpu.py: N ranks on ONE GPU): COLPHYS_BENCH_BACKEND=gloo swaps RCCL for
    # gloo (RCCL refuses two ranks on one device), COLPHYS_BENCH_SHARE_GPU=1 puts every rank on device 0.  The real
    # launch (one process per GPU, backend "nccl" = RCCL) sets neither.
    backend = os.environ.get("COLPHYS_BENCH_BACKEND", "nccl")
    if os.environ.get("COLPHYS_BENCH_SHARE_GPU", "0") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    params = abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, synth.limcnv_from_pref(pver))
    eng = ColumnPhysics(params, local_rank)
    c = synth.PHYSCONST
    d = eng.dev
    dt = synth.ZTODT
    pcols = 16

    # ---- resident inputs (physics_state of this rank's columns) and pre-allocated outputs -----------------
    S = {k: d(st[k]) for k in ("t", "q", "u", "v", "pint", "pmid", "pdel", "rpdel", "lnpint", "lnpmid", "pdeldry", "phis",
                               "pblh", "tpert", "landfrac", "cld", "rair", "zvir", "zi", "zm")}
    S["ps"] = d(st["pint"][:, -1].copy())
    zm_out = eng.alloc_zm_conv_tend_out(ncol, pver, pcols)
    if with_aero:
        eng.aero_tables_set(props, c)
        A = {k: d(aer[k]) for k in ("specmmr", "dgnumwet", "qaerwat")}
        sw_out = {k: eng.empty(ncol, synth.NSWBANDS, pver + 1) for k in ("tauxar", "wa", "ga", "fa")}
        sw_out["status"] = eng.empty(ncol, dtype=torch.int32)
        lw_out = dict(tauxar=eng.empty(ncol, synth.NLWBANDS, pver), status=eng.empty(ncol, dtype=torch.int32))
        if sw_diag:
            A["troplevchem"] = d(aer["troplevchem"]).to(torch.int32)
            swd_vis = eng.empty(3, len(props["nspec"]), ncol, pver)     # visible-band optics per mode, SW kernel -> diagnostics
            swd_out = dict(lev=eng.empty(abi.SWD_NLEV, ncol, pver), col=eng.empty(abi.SWD_NCOL + abi.SWD_NMODEF * len(props["nspec"]), ncol))

    # The optics kernels share no data with the ZM chain, so they go on a second HIP stream and fill the SIMDs
    # the latency-bound ZM kernels leave idle (760 column groups, then ~80, on 1 024 SIMDs); --serial disables it.
    side = torch.cuda.Stream(priority=0) if (with_aero and not args.serial) else None
    if with_aero:
        # kernel family of the LW optics (include/colphys.h colphys_aero_kernel_select; A/Bs in DESIGN.md section 4).  The same
        # kernels run in the --serial passes the profiles are taken from.
        eng.aero_kernel_select("auto", args.lw_kernel)

    def aero():
        eng.modal_aero_sw(S["pdeldry"], A["specmmr"], A["dgnumwet"], A["qaerwat"], out=sw_out, vis=swd_vis if sw_diag else None)
        if sw_diag:                                 # what the reference's modal_aero_sw also computes, for outfld
            eng.modal_aero_sw_diag(S["pdeldry"], S["pmid"], S["t"], A["specmmr"], A["dgnumwet"], A["qaerwat"], A["troplevchem"],
                                   sw_out["tauxar"], out=swd_out, vis=swd_vis)
        eng.modal_aero_lw(S["pdeldry"], A["specmmr"], A["dgnumwet"], A["qaerwat"], out=lw_out)

    step_start = torch.cuda.Event() if side is not None else None

    def substep():
        if side is not None:
            step_start.record()
        # zm_conv_tend (zm_conv_intr.F90:441-905) on resident data, ONE call of the C ABI: zm_convr -> physics_update ->
        # geopotential_t -> zm_conv_evap -> physics_update -> geopotential_t -> momtran -> physics_update -> geopotential_t,
        # ptend_all and the diagnostics accumulated on the device.  The engine enqueues the trigger first; the optics are
        # enqueued right behind it on the side stream so that its long-running waves are placed before the optics' blocks.
        eng.zm_conv_tend(S, dt, pcols=pcols, out=zm_out)
        if side is not None:
            side.wait_event(step_start)
            with torch.cuda.stream(side):
                aero()
            torch.cuda.current_stream().wait_stream(side)
        elif with_aero:
            aero()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The ~20 launches of one substep are captured once into a HIP graph and replayed (launch-bound otherwise:
    # the host-side ctypes calls alone cost ~1 ms per step); --no-graph replays the eager path.
    run_step = substep
    graph = None
    if not args.no_graph:
        substep()                                   # allocates the engine's workspaces before capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(priority=-1)        # the ZM chain is the critical path: high-priority queue
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            with torch.cuda.graph(graph, stream=cap):
                substep()
        torch.cuda.current_stream().wait_stream(cap)
        run_step = graph.replay

    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
    cols = torch.tensor([float(ncol)], dtype=torch.float64, device=eng.device)
    if world > 1:
        if backend != "nccl":                       # gloo reduces host tensors
            el, cols = el.cpu(), cols.cpu()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cols, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_cols = float(cols.item())

    # the replayed graph must leave exactly what the eager path leaves (outside the timed region)
    if graph is not None:
        def fingerprint():
            torch.cuda.synchronize()
            parts = [zm_out["ptend_s"].sum(), zm_out["prec"].sum(), zm_out["evapcdp"].sum(), zm_out["ptend_u"].sum(), zm_out["zm1"].sum()]
            if with_aero:
                parts += [sw_out["tauxar"].sum(), lw_out["tauxar"].sum()]
                if sw_diag:
                    parts += [swd_out["col"].sum(), swd_out["lev"].sum()]
            return torch.stack(parts).cpu()
        fp_graph = fingerprint()
        substep()
        if not torch.equal(fp_graph, fingerprint()):
            raise SystemExit("graph replay and eager launches disagree")

    # ---- per-kernel durations: HIP events on the launch stream, in a separate short pass ------------------
    eng.profile_enable(True)
    eng.profile_reset()
    side_saved, side = side, None              # per-kernel durations are measured un-overlapped
    for _ in range(max(3, min(args.steps, 10))):
        substep()
    side = side_saved
    torch.cuda.synchronize()
    prof = eng.profile()
    # ... and once more with the two streams overlapping (eager launches): what the same kernels take while they share the GPU
    prof_ov = None
    if side is not None:
        eng.profile_reset()
        for _ in range(3):
            substep()
        torch.cuda.synchronize()
        prof_ov = eng.profile()
    eng.profile_enable(False)
    ndeep = int(zm_out["lengath"].sum().item())
    status_bad = int((zm_out["status"] != 0).sum().item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cols * args.steps / elapsed
        kern_ms = {k: v[0] for k, v in prof.items() if v[1] > 0}
        per_call = {"geopotential_t": BYTES["geopotential_t"](pver), "zm_trigger": BYTES["zm_convr"](pver),
                    "zm_conv_evap": BYTES["zm_conv_evap"](pver), "momtran": BYTES["momtran"](pver),
                    # SURVEY.md 8d's figure.  With the diagnostics the launch also writes each mode's visible-band pext, pabs, pasm
                    # for the diagnostics kernel (3 x nmodes doubles per cell): an intermediate of this design, reported beside
                    # the algorithmic bytes (handover_bytes_per_column), not counted in them
                    "modal_aero_sw": BYTES["modal_aero_sw"](pver),
                    "modal_aero_lw": BYTES["modal_aero_lw"](pver)}
        # share of one substep: geopotential_t and physics_update run three times
        share = {k: v * (3 if k in ("geopotential_t", "physics_update") else 1) for k, v in kern_ms.items()}
        dom = max((k for k in share if k in per_call), key=lambda k: share[k])
        units = ndeep if dom == "momtran" else ncol
        ach = per_call[dom] * units / (kern_ms[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None, "traffic_unit": "GB per launch",
                    "algorithmic_gb_per_launch": round(per_call[dom] * units / 1e9, 4),
                    "avg_launch_ms": round(kern_ms[dom], 4), "bytes_per_column": per_call[dom], "columns_per_launch": units,
                    "kernel_ms": {k: round(v, 4) for k, v in kern_ms.items()},
                    "handover_bytes_per_column": (3 * 4 * pver * 8 if (sw_diag and dom == "modal_aero_sw") else 0),
                    "substep_bytes_per_column": substep_bytes(pver, with_aero and sw_diag),
                    "substep_hbm_frac": round(substep_bytes(pver, with_aero and sw_diag) * value / world / 1e9 / HBM_PEAK_GBS, 6),
                    "note": "path is fp64-transcendental / latency bound, not HBM bound (SURVEY.md 0-4, 8d)",
                    "duration_source": "HIP events on the launch stream in a separate un-overlapped pass after the timed region; "
                                       "rocprofv3 of `bench.py --serial --no-graph` averages the same launches "
                                       "(profiles/r02_kernel_stats_serial.csv); the default command's rocprofv3 average "
                                       "(profiles/r02_kernel_stats.csv) also contains the overlapped graph replays"}
        if args.ncol == 48602 and pver == 32:
            vi = valu_issue_of_substep(ms_per_step, with_aero)
            if vi is not None:
                roofline["valu_issue"] = vi
        if prof_ov is not None:
            roofline["kernel_ms_overlapped"] = {k: round(v[0], 4) for k, v in prof_ov.items() if v[1] > 0}
        pm = pmc_of(dom) if (args.ncol == 48602 and pver == 32) else None
        if pm is not None:
            roofline["traffic"], roofline["traffic_source"] = pm["traffic"], pm["source"]
            if "valu" in pm:
                roofline["valu_simd_util"] = pm["valu"]
        out = {
            "metric": baseline_metric(),
            "value": round(value, 1), "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("ZM (zm_convr + zm_conv_evap + momtran) + 3x geopotential_t"
                                    + ((" + modal_aero_sw (with its history diagnostics)" if sw_diag else " + modal_aero_sw (radiative outputs only)")
                                       + " + modal_aero_lw" if with_aero else "")
                                    + f", {args.ncol} columns x L{pver} per " + ("node" if args.strong else "GPU")),
                       "columns_per_rank": ncol, "pver": pver, "pcols": pcols, "deep_fraction": round(ndeep / ncol, 4),
                       "parallelism": f"static column split x{world}, no collectives on the data path",
                       "streams": 1 if side is None else 2, "hip_graph": graph is not None,
                       "columns_with_nonzero_status": status_bad},
            "roofline": roofline,
        }
        if cpu is not None:
            out["cpu_baseline"] = {"value": round(cpu["value"], 1), "unit": cpu["unit"], "cores": cpu["cores"],
                                   "kind": cpu["kind"], "sample": cpu["sample"], "seconds": round(cpu["seconds"], 3)}
            if cpu_port is not None:
                out["cpu_baseline_port"] = {"value": round(cpu_port["value"], 1), "unit": cpu_port["unit"], "cores": cpu_port["cores"],
                                            "kind": "port", "sample": cpu_port["sample"], "seconds": round(cpu_port["seconds"], 3)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
