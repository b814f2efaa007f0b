#!/usr/bin/env python3
"""Benchmark of the hot path: columns/sec for one "ZM + geopotential + aerosol-optics" physics substep on
synthetic ne30 x L32-shaped input (BASELINE.json: 48 602 columns x 32 levels), fp64.

One step = one pass of the substep over this rank's columns, inputs already resident in HBM
(cam_nor_physics_amd/substep.py -- the same object the parity tests drive):

    zm_convr -> physics_update -> geopotential_t -> zm_conv_evap -> physics_update -> geopotential_t
             -> momtran -> physics_update -> geopotential_t        ||        modal_aero_sw (+ diagnostics) -> modal_aero_lw

(the call order of zm_conv_tend, zm_conv_intr.F90:662-836, beside the two optics calls radiation makes,
modal_aer_opt.F90:485,1216).  Algorithmic bytes of the substep: 52 128 B/column at L32.

    python bench.py [--gpus N --steps K --warmup W]
        N = 1: runs in this process.  N > 1 without a launcher (WORLD_SIZE unset): this process starts
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`
        as a CHILD before it touches the GPU, relays its output and exits with its code.
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...        (what the driver runs)

Multi-GPU: one process per GPU, static chunk-aligned column split, NO collective on the data path; the
only communication is the barrier / max-reduce that brackets the timed region.  The headline (`value`, `scaling`) is
BASELINE.json's metric: STRONG scaling, ONE 48 602-column slab split over the N ranks.  The weak figure (every rank owns
a full slab) is measured in the same run, after the headline pass, and printed beside it (`weak`); `--weak` swaps the two.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on its launch stream), `cpu_baseline`
(the reference Fortran, or the C restatement if oracle/_ref is absent, on the host cores) and, at N = 1, `dropin`
(the same substep entered through the host-layout face: upload + device re-layout + kernels + download, per rank).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
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


def substep_bytes(L, sw_diag=True, with_aero=True):
    b = BYTES["zm_convr"](L) + BYTES["zm_conv_evap"](L) + BYTES["momtran"](L) + 3 * BYTES["geopotential_t"](L)
    if with_aero:
        b += BYTES["modal_aero_sw"](L) + BYTES["modal_aero_lw"](L) + (BYTES["modal_aero_sw_diag"](L) if sw_diag else 0)
    return b


PMC_KERNEL = {"geopotential_t": "geopotential_kernel", "zm_trigger": "zm_trigger_kernel", "zm_plume1": "zm_plume1_kernel",
              "zm_ascent2": "zm_ascent2_kernel", "zm_plume2": "zm_plume2_kernel",
              "zm_conv_evap": "zm_conv_evap_kernel", "momtran": "momtran_wave_kernel", "modal_aero_sw": "modal_aero_reg_kernel<true,",
              "modal_aero_lw": "modal_aero_reg_kernel<false,", "modal_aero_sw_diag": "modal_aero_sw_diag_vis_kernel"}


def kernel_source_hash():
    """sha256 over the kernel sources: a PMC summary is only quoted for the code it was taken from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cam-nor-physics_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_file():
    """newest profiles/*_pmc_summary.csv whose stamp (profiles/<same>.json: {"kernel_source_sha16": ...}, written by
    scripts/pmc_stamp.py when the passes are taken) names the kernel sources of THIS tree; (path, None) or (None, reason)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")))
    if not files:
        return None, "no profiles/*_pmc_summary.csv"
    f = files[-1]
    stamp = f[:-4] + ".json"
    if not os.path.exists(stamp):
        return None, f"{os.path.basename(f)} carries no source stamp"
    try:
        with open(stamp) as fh:
            sha = json.load(fh).get("kernel_source_sha16")
    except Exception:
        return None, f"{os.path.basename(stamp)} unreadable"
    if sha != kernel_source_hash():
        return None, f"{os.path.basename(f)} was taken from other kernel sources ({sha} != {kernel_source_hash()}): stale, not quoted"
    return f, None


def pmc_of(kernel, path):
    """PMC-derived figures of `kernel` from the committed passes of this same command (scripts/pmc_passes.sh ->
    profiles/*_pmc_summary.csv), or None:
      traffic  HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE reads half the bytes of a coalesced
               stream on gfx950 (MI355X_MICROARCH.md, HBM); calibrated here on physics_update_kernel, whose byte count
               is known exactly;
      valu     SIMD VALU utilisation = SQ_ACTIVE_INST_VALU * 4 cycles / (kernel cycles * 1 024 SIMDs): what bounds this
               path (fp64 vector ALU), reported beside the HBM fraction the contract asks for."""
    import csv
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith(PMC_KERNEL.get(kernel, "?")) and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
            out = {"traffic": round((2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024 / 1e9, 4),
                   "source": os.path.basename(path)}
            if r.get("SQ_ACTIVE_INST_VALU") and r.get("GRBM_GUI_ACTIVE"):
                out["valu"] = round(float(r["SQ_ACTIVE_INST_VALU"]) * 4.0 / (float(r["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0), 3)
            return out
    return None


def valu_issue_of_substep(ms_per_step, with_aero, path):
    """Substep-level fp64 VALU issue figure from the committed PMC passes: sum over the substep's launches of the wave-level
    VALU instructions (SQ_INSTS_VALU) x 4 cycles (one fp64 instruction per SIMD per 4 cycles, scripts/ubench) over
    1 024 SIMDs at the 2.4 GHz peak clock, as a fraction of the measured step time -- what bounds this path beside the HBM
    fraction the contract asks for."""
    import csv
    mult = {"geopotential_kernel": 3, "physics_update_kernel": 3}
    total = 0.0
    for r in csv.DictReader(open(path)):
        k = r["kernel"]
        if not r.get("SQ_INSTS_VALU") or (not with_aero and k.startswith("modal_aero")):
            continue
        total += float(r["SQ_INSTS_VALU"]) * mult.get(k, 1)
    ms_bound = total * 4.0 / (1024 * 2.4e9) * 1e3
    return {"wave_valu_instructions": round(total), "ms_at_full_issue": round(ms_bound, 4), "frac_of_step": round(ms_bound / ms_per_step, 4),
            "source": os.path.basename(path)}


def baseline_metric():
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "columns/sec, ZM+aer-opt physics substep, 48602 cols x L32 at 1/2/4/8 MI355X"


def host_info():
    """what the CPU baseline ran on: model, the machine's core count, the cores this process may use"""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        pass
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_cpu_quota": quota, "usable_cores": usable}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ncol", type=int, default=48602)
    ap.add_argument("--pver", type=int, default=32)
    ap.add_argument("--strong", action="store_true", help="(default) headline = ONE --ncol slab split over the ranks")
    ap.add_argument("--weak", action="store_true", help="headline = every rank owns a full --ncol slab; the strong figure goes beside it")
    ap.add_argument("--no-second-pass", action="store_true", help="N > 1: only the headline pass")
    ap.add_argument("--moist-boost", type=float, default=1.1,
                    help="synth.make_state(moist_boost=...): 1.1 puts the convecting fraction inside SURVEY.md 8d's 10-30 %% band "
                         "(18 %% at L32, 13 %% at L70); 1.0 is the rounds 1-2 workload (9.8 %% / 7 %%)")
    ap.add_argument("--no-aero", action="store_true", help="configs[1]: ZM only (zm_convr + zm_conv_evap + momtran)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-layout (drop-in face) measurement")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying a captured HIP graph")
    ap.add_argument("--no-sw-diag", action="store_true",
                    help="leave out modal_aero_sw's history diagnostics (modal_aer_opt.F90:896-993, 1046-1209), which the "
                         "reference computes inside modal_aero_sw on every call: radiative outputs only")
    ap.add_argument("--lw-kernel", choices=["auto", "bands-outer", "grouped"], default="auto",
                    help="kernel family of modal_aero_lw (colphys_aero_kernel_select).  auto = bands-outer: with the SW diagnostics "
                         "in the substep the optics stream ends last and the kernel that is faster alone wins (2.425 against 2.453 ms); "
                         "with --no-sw-diag `grouped` measures 0.6 %% faster")
    ap.add_argument("--lw-first", action="store_true", help="A/B: modal_aero_lw ahead of modal_aero_sw on the optics stream")
    ap.add_argument("--serial", action="store_true", help="all kernels on one stream (no ZM || optics overlap)")
    return ap.parse_args(argv)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child job.  Nothing in THIS process has
    touched the GPU (torch is not even imported yet) -- a process that has must never exec or fork GPU work."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, cwd=ROOT)
    return r.returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from cam_nor_physics_amd import synth
    from cam_nor_physics_amd.dist import shard_range

    pver = args.pver
    with_aero = not args.no_aero
    sw_diag = with_aero and not args.no_sw_diag
    headline = "weak" if args.weak else "strong"
    props = synth.make_aero_props(synth.DEFAULT_SEED) if with_aero else None

    def workload(mode):
        """this rank's columns: a chunk-aligned share of ONE slab (strong) or a slab of its own (weak)"""
        if mode == "strong":
            lo, hi = shard_range(args.ncol, world, rank)
            st_all = synth.make_state(args.ncol, pver, synth.DEFAULT_SEED, moist_boost=args.moist_boost)
            st = {k: (v[lo:hi] if hasattr(v, "shape") and v.shape[:1] == (args.ncol,) else v) for k, v in st_all.items()}
            st["ncol"] = hi - lo
            aer = None
            if with_aero:
                aer_all = synth.make_aero_state(args.ncol, pver, props, synth.DEFAULT_SEED)
                aer = {k: (v[lo:hi] if k == "troplevchem" else v[..., lo:hi, :]) for k, v in aer_all.items()}
            return st, aer
        st = synth.make_state(args.ncol, pver, synth.DEFAULT_SEED + rank, moist_boost=args.moist_boost)
        return st, (synth.make_aero_state(args.ncol, pver, props, synth.DEFAULT_SEED + rank) if with_aero else None)

    st, aer = workload(headline)
    ncol = st["ncol"]

    # ---- CPU baseline: rank 0 at N = 1 only, BEFORE this process touches the GPU (workers are forked) ----
    cpu = cpu_port = None
    host = host_info()
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from cpu_substep import run_cpu_baseline
        cpu = run_cpu_baseline(st, aer, props, pver, with_aero=with_aero, cores=host["usable_cores"])
        # the C restatement on the same cores as well: oracle/_ref is the reference's own Fortran, but built against
        # stand-in modules for the parent model it does not contain, so both CPU figures are reported
        cpu_port = (run_cpu_baseline(st, aer, props, pver, with_aero=with_aero, kind="port", cores=host["usable_cores"])
                    if cpu and cpu["kind"] == "reference" else None)

    import torch
    import torch.distributed as dist
    from cam_nor_physics_amd import abi
    from cam_nor_physics_amd.engine import ColumnPhysics
    from cam_nor_physics_amd.substep import ResidentSubstep

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
    pcols = 16

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_pair(elapsed, cols):
        el = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        cl = torch.tensor([float(cols)], dtype=torch.float64, device=eng.device)
        if world > 1:
            if backend != "nccl":                   # gloo reduces host tensors
                el, cl = el.cpu(), cl.cpu()
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            dist.all_reduce(cl, op=dist.ReduceOp.SUM)
        return float(el.item()), float(cl.item())

    def gather_ints(v):
        if world == 1:
            return [int(v)]
        t = torch.zeros(world, dtype=torch.float64, device=eng.device)
        t[rank] = float(v)
        if backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [int(x) for x in t.tolist()]

    def timed_pass(ss):
        """W untimed + EXACTLY K timed steps between barrier + synchronize pairs; max over ranks; whole-job columns"""
        if not args.no_graph:
            ss.capture()
        for _ in range(args.warmup):
            ss.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ss.step()
        barrier()
        elapsed, total_cols = reduce_pair(time.perf_counter() - t0, ss.ncol)
        # the replayed graph must leave exactly what the eager path leaves (outside the timed region), bit for bit
        if ss.graph is not None:
            fp_graph = ss.fingerprint()
            ss.clear_outputs()
            ss.launch()
            if not torch.equal(fp_graph, ss.fingerprint()):
                raise SystemExit("graph replay and eager launches disagree")
        return elapsed, total_cols

    ss = ResidentSubstep(eng, st, aer, props, pcols=pcols, sw_diag=sw_diag, two_streams=not args.serial, lw_kernel=args.lw_kernel, lw_first=args.lw_first)
    elapsed, total_cols = timed_pass(ss)

    # ---- per-kernel durations: HIP events on the launch stream, in a separate short pass ------------------
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(max(3, min(args.steps, 10))):
        ss.launch(overlap=False)                    # per-kernel durations are measured un-overlapped
    torch.cuda.synchronize()
    prof = eng.profile()
    # ... and once more with the two streams overlapping (eager launches): what the same kernels take while they share the GPU
    prof_ov = None
    if ss.side is not None:
        eng.profile_reset()
        for _ in range(3):
            ss.launch()
        torch.cuda.synchronize()
        prof_ov = eng.profile()
    eng.profile_enable(False)
    ndeep = int(ss.zm_out["lengath"].sum().item())
    status_bad = int((ss.zm_out["status"] != 0).sum().item())
    cols_per_rank = gather_ints(ncol)

    # ---- N > 1: the other scaling mode, after the headline (one live graph at a time) -------------------------
    second = None
    if world > 1 and not args.no_second_pass:
        ss.release_graph()
        del ss
        torch.cuda.empty_cache()
        other = "strong" if headline == "weak" else "weak"
        st2, aer2 = workload(other)
        ss2 = ResidentSubstep(eng, st2, aer2, props, pcols=pcols, sw_diag=sw_diag, two_streams=not args.serial, lw_kernel=args.lw_kernel, lw_first=args.lw_first)
        e2, c2 = timed_pass(ss2)
        second = {"scaling": other, "value": round(c2 * args.steps / e2, 1), "unit": "columns/s", "ms_per_step": round(e2 / args.steps * 1e3, 4),
                  "columns_per_rank": gather_ints(ss2.ncol), "total_columns": int(c2)}
        ss2.release_graph()
        del ss2
    elif world == 1:
        ss.release_graph()

    # ---- N = 1: the drop-in face -- host arrays in the reference's (pcols,pver,nchunks) layout in, results out -------------
    dropin = None
    if world == 1 and not args.no_dropin:
        try:
            from cam_nor_physics_amd.hostface import measure_dropin
            dropin = measure_dropin(eng, st, aer, props, pcols=pcols, sw_diag=sw_diag, resident_ms=elapsed / args.steps * 1e3)
        except ImportError:
            dropin = None

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
        sb = substep_bytes(pver, sw_diag, with_aero)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None, "traffic_unit": "GB per launch",
                    "algorithmic_gb_per_launch": round(per_call[dom] * units / 1e9, 4),
                    "avg_launch_ms": round(kern_ms[dom], 4), "bytes_per_column": per_call[dom], "columns_per_launch": units,
                    "kernel_ms": {k: round(v, 4) for k, v in kern_ms.items()},
                    "handover_bytes_per_column": (3 * 4 * pver * 8 if (sw_diag and dom == "modal_aero_sw") else 0),
                    "substep_bytes_per_column": sb,
                    "substep_hbm_frac": round(sb * value / world / 1e9 / HBM_PEAK_GBS, 6),
                    "note": "path is fp64-transcendental / latency bound, not HBM bound (SURVEY.md 0-4, 8d)",
                    "duration_source": "HIP events on the launch stream in a separate un-overlapped pass after the timed region; "
                                       "rocprofv3 of `bench.py --serial --no-graph` averages the same launches "
                                       "(profiles/r03_kernel_stats_serial.csv); the default command's rocprofv3 average "
                                       "(profiles/r03_kernel_stats.csv) also contains the overlapped graph replays"}
        if prof_ov is not None:
            roofline["kernel_ms_overlapped"] = {k: round(v[0], 4) for k, v in prof_ov.items() if v[1] > 0}
        # PMC figures come from committed passes of this very command and are quoted only for the default workload AND only
        # while the summary's stamp names the kernel sources of this tree; otherwise traffic stays null and says why
        default_workload = (args.ncol == 48602 and pver == 32 and world == 1 and args.moist_boost == 1.1 and with_aero and sw_diag)
        pmc_path, why = pmc_file()
        if not default_workload:
            roofline["traffic_note"] = "PMC passes are taken on the default N = 1 workload only"
        elif pmc_path is None:
            roofline["traffic_note"] = why
        else:
            pm = pmc_of(dom, pmc_path)
            if pm is not None:
                roofline["traffic"], roofline["traffic_source"] = pm["traffic"], pm["source"]
                if "valu" in pm:
                    roofline["valu_simd_util"] = pm["valu"]
            vi = valu_issue_of_substep(ms_per_step, with_aero, pmc_path)
            if vi is not None:
                roofline["valu_issue"] = vi
        slab = f"{args.ncol} columns x L{pver}"
        out = {
            "metric": baseline_metric(),
            "value": round(value, 1), "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": headline, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("ZM (zm_convr + zm_conv_evap + momtran) + 3x geopotential_t"
                                    + ((" + modal_aero_sw (with its history diagnostics)" if sw_diag else " + modal_aero_sw (radiative outputs only)")
                                       + " + modal_aero_lw" if with_aero else "")
                                    + ", " + (f"ONE slab of {slab} split over {world} GPU(s)" if headline == "strong" else f"{slab} per GPU")),
                       "columns_per_rank": ncol, "columns_of_every_rank": cols_per_rank, "total_columns": int(total_cols),
                       "pver": pver, "pcols": pcols, "deep_fraction": round(ndeep / ncol, 4), "moist_boost": args.moist_boost,
                       "parallelism": f"static chunk-aligned column split x{world}, no collectives on the data path",
                       "world_size_seen": world, "backend": (backend if world > 1 else None),
                       "streams": 1 if (args.serial or not with_aero) else 2, "hip_graph": not args.no_graph,
                       "columns_with_nonzero_status": status_bad},
            "roofline": roofline,
            "host": host,
        }
        this = {"value": out["value"], "unit": "columns/s", "ms_per_step": out["ms_per_step"], "columns_per_rank": cols_per_rank,
                "total_columns": int(total_cols)}
        out[headline] = this
        if second is not None:
            out[second.pop("scaling")] = second
        elif world == 1:
            out["weak" if headline == "strong" else "strong"] = dict(this, note="N = 1: the two scaling modes are the same run")
        if dropin is not None:
            out["dropin"] = dropin
        if cpu is not None:
            out["cpu_baseline"] = {"value": round(cpu["value"], 1), "unit": cpu["unit"], "cores": cpu["cores"],
                                   "kind": cpu["kind"], "sample": cpu["sample"], "seconds": round(cpu["seconds"], 3),
                                   "cpu_model": host["cpu_model"], "cores_note": "every core this process may use (min of affinity and cgroup quota)"}
            if cpu_port is not None:
                out["cpu_baseline_port"] = {"value": round(cpu_port["value"], 1), "unit": cpu_port["unit"], "cores": cpu_port["cores"],
                                            "kind": "port", "sample": cpu_port["sample"], "seconds": round(cpu_port["seconds"], 3)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
