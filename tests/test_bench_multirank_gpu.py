"""Rehearsal of bench.py's multi-rank path on ONE GPU: two ranks under torch.distributed.run, gloo instead of RCCL (RCCL
refuses two ranks on one device), both ranks on device 0.  Everything else is the code the driver runs at N = 2, 4, 8:
rank / world from the environment, per-rank slabs, barrier + max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("strong", [False, True])
def test_two_ranks_on_one_gpu(strong):
    env = dict(os.environ, COLPHYS_BENCH_BACKEND="gloo", COLPHYS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ncol = 4003
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--ncol", str(ncol), "--no-cpu-baseline"] + (["--strong"] if strong else [])
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == ("strong" if strong else "weak")
    total = ncol if strong else 2 * ncol
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3          # whole-job columns / max-over-ranks time
    from cam_nor_physics_amd.dist import shard_range
    lo, hi = shard_range(ncol, 2, 0)
    assert d["config"]["columns_per_rank"] == (hi - lo if strong else ncol)
    assert d["config"]["columns_with_nonzero_status"] == 0
