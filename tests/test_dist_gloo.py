"""N > 1 path on CPU: two gloo ranks run bench.py's rank logic (static column split, no data-path collective,
barrier + max-over-ranks timing) with the C oracle standing in for the device step, and the union of their
results equals the single-rank run."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ncol, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import time

    import torch.distributed as dist

    import cases
    from c_oracle import COracle
    from cam_nor_physics_amd import synth
    from cam_nor_physics_amd.dist import shard_range, timed_region_max

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = synth.make_state(ncol, 32, 99)
    lo, hi = shard_range(ncol, world, rank)
    args = cases.zm_args_from_state(st, st["zi"], st["zm"])
    mine = tuple(a[lo:hi] if isinstance(a, np.ndarray) else a for a in args)
    co = COracle(cases.default_params(32))
    dist.barrier()
    t0 = time.perf_counter()
    o = co.zm_convr(*mine)
    dist.barrier()
    tmax = timed_region_max(time.perf_counter() - t0)
    q.put((rank, lo, hi, o["heat"], o["prec"], o["lengath"], o["ideep"], tmax))
    dist.destroy_process_group()


def test_two_ranks_static_split_matches_single_rank():
    ncol, world = 1000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ncol, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import cases
    from c_oracle import COracle
    from cam_nor_physics_amd import synth
    st = synth.make_state(ncol, 32, 99)
    full = COracle(cases.default_params(32)).zm_convr(*cases.zm_args_from_state(st, st["zi"], st["zm"]))
    assert res[0][1] == 0 and res[-1][2] == ncol and res[0][2] == res[1][1]      # disjoint cover
    heat = np.concatenate([r[3] for r in res]); prec = np.concatenate([r[4] for r in res])
    assert np.array_equal(heat, full["heat"]) and np.array_equal(prec, full["prec"])
    # chunk-aligned split => the per-chunk gathered layout of the shards concatenates to the global one
    assert np.array_equal(np.concatenate([r[5] for r in res]), full["lengath"])
    assert np.array_equal(np.concatenate([r[6] for r in res]), full["ideep"])
    assert res[0][7] == res[1][7] > 0                                             # both ranks report the same max time
