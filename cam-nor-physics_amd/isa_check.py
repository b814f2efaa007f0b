"""ISA check of the built kernels for a known AMDGPU backend miscompile (round 3; DESIGN.md "the plume hazard").

Under register pressure the register allocator's live-range-split copy (`v_mov_b32 vA, vB`) can land at the head of a
loop-exit block AHEAD of the `s_or_b64 exec, exec, s[..]` that restores the lanes the divergent loop had masked off.  The
loop is left when EXEC reaches 0, so the copy runs for no lane; vB is then reused, and a later `v_mov_b32 vC, vA` reads a
register nothing ever wrote.  In zm_plume2_kernel<2> (two columns per wave, 168-VGPR budget) the victim was the base offset
of one LDS array (sd, qd or md -- whichever the allocation of the day split there): non-deterministic garbage in q1q2's
reads of that array, appearing and vanishing with unrelated source changes, and first taken for an LDS ordering hazard.
Nothing in the source is wrong in such a build, so the build checks the ISA: a vector instruction in a block every entry of
which has EXEC = 0, ahead of that block's EXEC restore, is never intended by the backend."""
from __future__ import annotations

import os
import re
import subprocess
import tempfile

BLOCK = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:|[_A-Za-z][\w.$]*:)")
VECTOR = re.compile(r"^\s+(v_|ds_|global_|scratch_|buffer_|flat_)")
# vector instructions that ignore EXEC (SGPR spills to VGPR lanes)
EXEC_BLIND = re.compile(r"^\s+v_(readlane|writelane|readfirstlane)_b32")


def scan(asm_path: str):
    """A labelled block every way into which has EXEC = 0 -- reached by fall-through from a loop's closing
    `s_cbranch_execnz <header>` and / or by `s_cbranch_execz <label>` only -- must restore EXEC before its first vector
    instruction: one ahead of the restore executes for no lane at all, which the backend never intends."""
    with open(asm_path) as f:
        lines = f.read().split("\n")
    code = lambda l: bool(l.strip()) and not l.lstrip().startswith((";", ".")) or l.startswith(".LBB")
    # how each label is reached by explicit branches
    reached: dict[str, set[str]] = {}
    for l in lines:
        m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m:
            reached.setdefault(m.group(2), set()).add(m.group(1))
    hits = []
    kernel = None
    prev = ""                                 # previous instruction line
    i = 0
    while i < len(lines):
        l = lines[i]
        m = BLOCK.match(l)
        if m and not l.startswith((".LBB", "; %bb")):
            kernel = m.group(1).rstrip(":")
        if l.startswith(".LBB"):
            label = l.split(":")[0]
            fall_zero = re.match(r"^\s+s_cbranch_execnz", prev) is not None
            no_fall = re.match(r"^\s+(s_branch|s_endpgm|s_setpc)", prev) is not None
            explicit = reached.get(label, set())
            if (fall_zero or no_fall) and explicit <= {"s_cbranch_execz"} and (fall_zero or explicit):
                j = i + 1
                while j < len(lines):
                    x = lines[j]
                    if BLOCK.match(x) and x.startswith(".LBB"):
                        break
                    if re.match(r"^\s+s_\w+\s+exec\b", x) or re.match(r"^\s+s_\w*saveexec", x) or re.match(r"^\s+s_(cbranch|branch|endpgm|setpc)", x):
                        break
                    if VECTOR.match(x) and not EXEC_BLIND.match(x):
                        hits.append((kernel, j + 1, x.strip()))
                    j += 1
        if code(l):
            prev = l                           # a label directly above: the fall-through's EXEC is unknown, neither test matches
        i += 1
    return hits



def check_sources(extra_flags=(), sources=None):
    """Compiles the HIP sources to gfx950 assembly (device side only, ~20 s) and scans them; returns the list of hits
    (source, kernel, line, instruction)."""
    from .build import CSRC, FLAGS, HIPCC, SOURCES
    hits = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in (sources or SOURCES):
            out = os.path.join(tmp, src.replace(".hip", ".s"))
            subprocess.check_call([HIPCC, *FLAGS, *extra_flags, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                                  stderr=subprocess.DEVNULL)
            hits += [(src, *h) for h in scan(out)]
    return hits


def format_hits(hits):
    return "\n".join(f"{src}: {kernel}: line {no}: `{ins}` runs with EXEC = 0 (ahead of its block's EXEC restore)"
                     for src, kernel, no, ins in hits)
