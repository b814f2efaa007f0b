"""Builds libcolphys_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "libcolphys_hip.so")
SOURCES = ["colphys_api.hip", "k_zm_convr.hip", "k_scans.hip", "k_modal_aero.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the parity path keeps the reference's operation order (no FMA contraction)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-value", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # .hip, .h and .inc alike
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "colphys.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """extra_flags / out: experiment builds (A/B variants of a kernel) next to the shipped library"""
    if out is None and not force and not _stale():
        return LIB
    os.makedirs(BUILD, exist_ok=True)
    objs = []
    tag = "" if out is None else "." + os.path.basename(out).replace(".so", "")
    for src in SOURCES:
        obj = os.path.join(BUILD, src.replace(".hip", tag + ".o"))
        cmd = [HIPCC, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    # the same flags, device side to assembly: refuse a build the backend miscompiled (isa_check.py; ~20 s)
    from . import isa_check
    hits = isa_check.check_sources(extra_flags)
    if hits and out is None:
        raise RuntimeError("hipcc left vector instructions in front of a loop-exit block's EXEC restore (they run with EXEC = 0):\n"
                           + isa_check.format_hits(hits) + "\nno library was linked; see cam_nor_physics_amd/isa_check.py")
    if hits:
        print("isa_check (experiment build, linked anyway):\n" + isa_check.format_hits(hits))
    lib = out or LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if out is not None:                 # an experiment build: its objects are of no further use (and would travel with gpurun)
        for o in objs:
            os.remove(o)
    return lib


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
