"""Rehearsal of bench.py's multi-rank path on ONE GPU.  The STANDALONE command `python bench.py --gpus 2` is what runs: the
parent starts two ranks under torch.distributed.run as a child job before it touches the GPU; gloo instead of RCCL (RCCL
refuses two ranks on one device), both ranks on device 0.  Everything else is the code the driver runs at N = 2, 4, 8:
rank / world from the environment, static chunk-aligned split, barrier + max-over-ranks timing, one JSON line from rank 0
carrying the strong figure (BASELINE.json's metric: ONE slab split over the ranks) and the weak one."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("headline", ["strong", "weak"])
def test_two_ranks_on_one_gpu_standalone_command(headline):
    env = dict(os.environ, COLPHYS_BENCH_BACKEND="gloo", COLPHYS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    ncol = 4003
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ncol", str(ncol),
           "--no-cpu-baseline"] + (["--weak"] if headline == "weak" else [])
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == headline
    assert d["config"]["world_size_seen"] == 2
    from cam_nor_physics_amd.dist import shard_range
    shares = [shard_range(ncol, 2, r_)[1] - shard_range(ncol, 2, r_)[0] for r_ in range(2)]
    assert sum(shares) == ncol
    for mode, cols, total in (("strong", shares, ncol), ("weak", [ncol, ncol], 2 * ncol)):
        s = d[mode]
        assert s["columns_per_rank"] == cols and s["total_columns"] == total
        assert abs(s["value"] - total / (s["ms_per_step"] * 1e-3)) / s["value"] < 1e-3      # whole-job columns / max-over-ranks time
    assert d["value"] == d[headline]["value"] and d["ms_per_step"] == d[headline]["ms_per_step"]
    assert d["config"]["columns_with_nonzero_status"] == 0


def test_failing_rank_fails_the_standalone_command():
    """a child that fails must fail the parent: --pver 3 is refused by the engine in every rank"""
    env = dict(os.environ, COLPHYS_BENCH_BACKEND="gloo", COLPHYS_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--ncol", "64",
                        "--pver", "3", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
