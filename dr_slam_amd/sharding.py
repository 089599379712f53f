"""Multi-GPU mode of the front-end: independent sequences sharded across ranks.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).  Frames
of different sequences are independent, so the steady state has NO collective: each rank owns
sequence `seed_base + rank`.  The only exchanges are (i) one broadcast of constant tables from rank 0
at start-up (the ORB vocabulary's slot once the BoW row lands; the scale tables until then) and
(ii) a MAX all-reduce of the elapsed time / SUM of frame counters for reporting.
"""
from __future__ import annotations

import numpy as np


def rank_seed(seed_base: int, rank: int) -> int:
    return seed_base + rank


def broadcast_tables(tables: np.ndarray, device, dist):
    """Rank 0's tables overwrite every rank's copy; returns the received array."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(tables)).to(device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, 0)
    return t.cpu().numpy()


def reduce_elapsed_and_frames(elapsed_s: float, frames: int, device, dist):
    """-> (max elapsed over ranks, total frames over ranks)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_s, frames
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([frames], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def pingpong_order(batch: int, n_distinct: int):
    """0,1,..,n-1,n-2,..,0,1,..: every adjacent pair of the batch is a real frame-to-frame motion."""
    order, k, d = [], 0, 1
    for _ in range(batch):
        order.append(k)
        if n_distinct > 1:
            if k + d < 0 or k + d >= n_distinct:
                d = -d
            k += d
    return order
