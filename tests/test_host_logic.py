"""CPU tests of the host-side logic: static column split, limcnv, synthetic generator determinism, tolerances."""
import numpy as np
import pytest

from cam_nor_physics_amd import synth
from cam_nor_physics_amd.dist import shard_range
from util import assert_close, gathered_mask, rel_err_profile


@pytest.mark.parametrize("ncol,world,pcols", [(48602, 8, 16), (48602, 1, 16), (777602, 8, 16), (100, 8, 16), (5, 8, 16), (48602, 3, 24)])
def test_shard_range_partitions_exactly(ncol, world, pcols):
    owned = np.zeros(ncol, dtype=np.int32)
    sizes = []
    for r in range(world):
        lo, hi = shard_range(ncol, world, r, pcols)
        assert 0 <= lo <= hi <= ncol
        assert lo % pcols == 0 or lo == hi == ncol  # whole chunks: the gathered layout stays local
        owned[lo:hi] += 1
        sizes.append(hi - lo)
    assert (owned == 1).all()
    assert max(sizes) - min(sizes) <= 2 * pcols or ncol < world * pcols


def test_shard_range_rejects_bad_rank():
    with pytest.raises(ValueError):
        shard_range(100, 4, 4)


def test_limcnv_is_first_interface_below_40hpa():
    for pver in (26, 32, 70):
        lim = synth.limcnv_from_pref(pver)
        pe = synth.PTOP + synth.sigma_edges(pver) * (1.0e5 - synth.PTOP)
        assert pe[lim - 1] < 4.0e3 <= pe[lim]      # 1-based interface index (zm_conv_intr.F90:356-368)


def test_synthetic_state_is_deterministic_and_physical():
    a = synth.make_state(200, 32, 3)
    b = synth.make_state(200, 32, 3)
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
    assert (np.diff(a["pint"], axis=1) > 0).all() and (a["q"] > 0).all() and (a["t"] >= 195).all()
    assert (np.diff(a["zi"], axis=1) < 0).all() and np.all(a["zi"][:, -1] == 0)
    assert np.allclose(a["pdel"], a["pint"][:, 1:] - a["pint"][:, :-1])


def test_tolerance_helpers():
    ref = np.array([[1.0, 0.0, -2.0], [1e-3, 1e-19, 0.0]])
    a = ref.copy(); a[1, 1] = 0.0                       # a cancellation residue must not fail the check
    assert rel_err_profile(a, ref) < 1e-10
    a[0, 0] += 1e-8
    with pytest.raises(AssertionError):
        assert_close("x", a, ref)
    with pytest.raises(AssertionError):
        assert_close("i", np.array([1, 2], dtype=np.int32), np.array([1, 3], dtype=np.int32))
    m = gathered_mask(np.array([2, 0, 1]), 40, 16)
    assert m.sum() == 3 and m[0] and m[1] and m[32] and not m[16]


def test_bench_host_info_and_pmc_stamp_logic(tmp_path, monkeypatch):
    """bench.py's host description and its refusal of a PMC summary taken from other kernel sources"""
    import json
    import bench
    h = bench.host_info()
    assert 1 <= h["usable_cores"] <= h["sched_affinity"] and isinstance(h["cpu_model"], str)
    sha = bench.kernel_source_hash()
    assert len(sha) == 16 and sha == bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: sha)
    assert bench.pmc_file()[0] is None
    (prof / "r09_pmc_summary.csv").write_text('kernel,FETCH_SIZE,WRITE_SIZE\n"modal_aero_reg_kernel<true, true, true>",1000,500\n')
    assert bench.pmc_file()[0] is None and "stamp" in bench.pmc_file()[1]
    (prof / "r09_pmc_summary.json").write_text(json.dumps({"kernel_source_sha16": "0" * 16}))
    assert bench.pmc_file()[0] is None and "stale" in bench.pmc_file()[1]
    (prof / "r09_pmc_summary.json").write_text(json.dumps({"kernel_source_sha16": sha}))
    path, why = bench.pmc_file()
    assert why is None and bench.pmc_of("modal_aero_sw", path)["traffic"] == round((2 * 1000 + 500) * 1024 / 1e9, 4)


def test_bench_parent_of_a_multi_gpu_run_spawns_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher hands over to torch.distributed.run as a child and returns its code"""
    import bench
    seen = {}

    def fake_run(cmd, env=None, cwd=None):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.spawn_ranks(4) == 7
    c = seen["cmd"]
    assert c[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in c and c[c.index("--nproc-per-node") + 1] == "4"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in seen["env"]
    assert "torch" not in [m for m in ("torch",) if m in bench.__dict__]       # the parent module itself never imported torch
