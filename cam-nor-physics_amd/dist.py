"""Static column split across the GPUs of one node: every grid column is independent, so each
rank owns a contiguous, chunk-aligned slice and no collective runs on the data path."""
from __future__ import annotations


def shard_range(ncol: int, world: int, rank: int, pcols: int = 16) -> tuple[int, int]:
    """[lo, hi) of rank's columns: whole chunks of `pcols` (so the per-chunk gathered layout stays local),
    chunk counts differing by at most one across ranks, every column owned exactly once."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    nchunks = (ncol + pcols - 1) // pcols
    base, extra = divmod(nchunks, world)
    c_lo = rank * base + min(rank, extra)
    c_hi = c_lo + base + (1 if rank < extra else 0)
    return min(c_lo * pcols, ncol), min(c_hi * pcols, ncol)


def timed_region_max(elapsed: float, group=None) -> float:
    """max over ranks of a per-rank wall time (what bench.py reports)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64)
    if dist.is_initialized():
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
