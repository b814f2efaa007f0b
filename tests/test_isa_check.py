"""The ISA check of the build (cam_nor_physics_amd/isa_check.py): the scanner on hand-written blocks, the shipped sources
(no hit), and the configuration that reproduces the backend's miscompile (round 3's "plume hazard") so the detector is
known to still see it."""
import os
import sys
import textwrap

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cam_nor_physics_amd import isa_check


def _scan(tmp_path, asm):
    p = tmp_path / "k.s"
    p.write_text(textwrap.dedent(asm))
    return isa_check.scan(str(p))


def test_copy_ahead_of_the_exec_restore_of_a_loop_exit_is_flagged(tmp_path):
    hits = _scan(tmp_path, """\
        kern:
        .LBB0_1:
        \tv_add_u32_e32 v1, 32, v1
        \ts_andn2_b64 exec, exec, s[14:15]
        \ts_cbranch_execnz .LBB0_1
        .LBB0_2:
        \ts_mov_b32 s8, 0
        \tv_mov_b32_e32 v156, v153
        \tv_readlane_b32 s4, v166, 3
        \ts_or_b64 exec, exec, s[0:1]
        \tv_mov_b32_e32 v153, v149
        \ts_endpgm
        """)
    assert [(k, ins) for k, _, ins in hits] == [("kern", "v_mov_b32_e32 v156, v153")]


def test_restore_first_and_other_entries_are_not_flagged(tmp_path):
    # the restore comes first; and a block that something reaches with lanes on (vccnz) proves nothing
    assert _scan(tmp_path, """\
        kern:
        .LBB0_1:
        \ts_andn2_b64 exec, exec, s[14:15]
        \ts_cbranch_execnz .LBB0_1
        .LBB0_2:
        \ts_or_b64 exec, exec, s[0:1]
        \tv_mov_b32_e32 v156, v153
        \ts_cbranch_vccnz .LBB0_3
        \ts_cbranch_execnz .LBB0_1
        .LBB0_3:
        \tv_mov_b32_e32 v1, v2
        \ts_or_b64 exec, exec, s[2:3]
        \ts_endpgm
        """) == []


def test_shipped_sources_are_clean():
    hits = isa_check.check_sources()
    assert hits == [], isa_check.format_hits(hits)


def test_detector_sees_the_miscompile_the_compiler_really_produced():
    # the block as hipcc emitted it for zm_plume2_kernel<2> (commit e8b916b, -DSERIAL_UNROLL=1; tests/golden/): the two split
    # copies ahead of `s_or_b64 exec, exec, s[0:1]` at the exit of the mass-flux scaling loop.  (The live reproducer went away
    # with the plume kernels' LDS re-packing -- another register allocation, no split there -- so the evidence is kept as data.)
    hits = isa_check.scan(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "isa_plume2_miscompile_excerpt.s"))
    assert [ins for _, _, ins in hits] == ["v_mov_b32_e32 v139, v159", "v_mov_b32_e32 v156, v153"]
