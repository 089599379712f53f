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


# ---------------------------------------------------------------------------------------------------
# Workloads of BASELINE.json's configs, one rank's share

def rank_workload(config: int, rank: int):
    """-> (seed, camera, scene kind, frames per sequence) of the sequence rank `rank` owns.

    config 2: the single-GPU headline (TUM3 intrinsics, room_boxes; every rank runs seed 10+rank so N ranks never
              render the same images);
    config 5: a 1280x960 RealSense-style stream (BASELINE.json configs[4]), 16 distinct frames per rank;
    config 4: eight independent 256-frame sequences, seeds 10..17, intrinsics cycling TUM1/TUM2/TUM3 yaml - two
              thirds of the ranks have lens distortion, so Frame::UndistortKeyPoints (src/Frame.cc:835-871) is live.
    """
    from . import synth
    if config == 4:
        cam = (synth.TUM1, synth.TUM2, synth.TUM3)[rank % 3]
        return 10 + (rank % 8), cam, "room_boxes", 256
    if config == 2:
        return rank_seed(10, rank), synth.TUM3, "room_boxes", 64
    if config == 5:   # 1280x960 RealSense-style stream (a parity / roofline case): D435 intrinsics x 2
        return rank_seed(20, rank), synth.REALSENSE.scaled(2.0), "corridor", 16
    raise ValueError(f"no sharded workload for config {config}")


def _render_one(args):
    from . import synth
    seed, k, cam, kind = args
    sc = synth.Scene(seed, kind)
    T = synth.pose_twc(k)
    g, d = synth.render(sc, cam, T, k)
    return g, d, T


def render_sequence(seed: int, n_frames: int, cam, kind: str = "room_boxes", workers: int = 1):
    """synth.sequence(seed, n_frames, cam, kind) as a list, rendered on `workers` processes (frames are independent).
    Call it BEFORE the process touches the GPU: the pool forks."""
    jobs = [(seed, k, cam, kind) for k in range(n_frames)]
    if workers <= 1 or n_frames < 4:
        return [_render_one(j) for j in jobs]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(workers, n_frames)) as pool:
        return pool.map(_render_one, jobs, chunksize=max(1, n_frames // (workers * 4)))


def host_cpus() -> int:
    """CPUs this process may use: the cgroup quota if there is one, else the affinity mask."""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n
