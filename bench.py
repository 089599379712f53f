#!/usr/bin/env python3
"""bench.py — RGB-D frames/sec (extract+match) at 640x480 on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of `--batch` synthetic frames of one sequence,
already resident in HBM: ORB extraction (pyramid, FAST, quadtree, blur, rBRIEF) -> stereo/grid glue ->
SearchByProjection(frame k, frame k-1) for every consecutive pair of the batch (BASELINE config 2).
N>1: one process per GPU (torch.distributed, backend nccl == RCCL), every rank runs its own sequence —
no data-path collective; timing is barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, HIP-event
timed) and `cpu_baseline` (the CPU oracle timed on this host, single thread, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak

# SURVEY.md §8(d) algorithmic bytes per 640x480 frame, by stage (N = 1000 keypoints)
ALGO_BYTES = {
    "pyramid": 307200 + 1158012 + 926546,     # read gray + write bordered pyramid + read for resize
    "fast": 950532,                           # FAST read of every level interior
    "blur": 1901064,                          # blur read + write
    "desc": 1922000 + 60000,                  # orientation + descriptor patches + outputs
    "match": 676000,                          # window match (a-10)
}


WORKLOAD_TEXT = {
    2: "640x480 synthetic RGB-D (TUM3 intrinsics), ORB extract 1000/1.2/8/20/7 + SearchByProjection(frame k, "
       "frame k-1, th=15) on consecutive frames; BASELINE config 2",
    4: "640x480 synthetic RGB-D, one 256-frame sequence per rank (seed 10+rank, intrinsics cycling TUM1/TUM2/TUM3: "
       "UndistortKeyPoints live on two thirds of the ranks), ORB extract 1000/1.2/8/20/7 + SearchByProjection(frame "
       "k, frame k-1, th=15); BASELINE config 4",
}


def make_batch(base, batch: int):
    """Ping-pong over the rendered sequence so every adjacent pair of the batch is a real
    frame-to-frame motion: 0,1,..,n-1,n-2,..,1,0,1,.."""
    from dr_slam_amd.sharding import pingpong_order
    order = pingpong_order(batch, len(base))
    gray = np.stack([base[i][0] for i in order])
    depth = np.stack([base[i][1] for i in order])
    Twc = np.stack([base[i][2] for i in order]).astype(np.float64)
    Tcw = np.linalg.inv(Twc)
    return gray, depth, Tcw.astype(np.float32), Twc.astype(np.float32)


def cpu_baseline(base, cam, budget_s: float = 12.0):
    """Oracle (kind 'port', 1 thread): extract + SearchByProjection against the previous frame."""
    from oracle import oracle as orc
    o = orc.OrbOracle()
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = np.float32(1.0) / np.float32(cam.depth_factor)
    prev = None
    n, t0 = 0, time.perf_counter()
    i = 0
    while True:
        g, d, Twc = base[i % len(base)]
        kps, desc = o(g)
        fo = orc.FrameOracle(kps, desc, orc.depth_to_float(d, inv), K4, cam.bf, cam.w, cam.h, o.scale,
                             dist=cam.dist)
        Tcw = np.linalg.inv(Twc).astype(np.float32)
        if prev is not None:
            pf, pTwc, pTcw = prev
            world, valid = pf.unproject(pTwc.astype(np.float32))
            mp = np.zeros(pf.N, orc.MAPPOINT_DTYPE)
            mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, pf.desc
            orc.search_by_projection_last(fo, pf, Tcw, pTcw, mp, 15.0, False, True)
        prev = (fo, Twc, Tcw)
        n += 1
        i += 1
        el = time.perf_counter() - t0
        if el > budget_s and n >= 8:
            break
    return {"value": n / el, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames 640x480 (extract + SearchByProjection vs previous frame), CPU oracle "
                      f"-O3 -march=x86-64-v3 -ffp-contract=off, 1 thread, {el:.1f} s"}


def launch(args) -> int:
    """`--gpus N` without a launcher around us: start N fresh rank processes (one per GPU) BEFORE this process makes
    any GPU call - the parent never initialises HIP, it only counts devices - and pass rank 0's JSON line through.
    Non-zero exit if any rank fails.  On a box with fewer than N devices the ranks share device 0 over gloo
    (rehearsal of the N-rank control path; flagged in the JSON line, not a scaling measurement)."""
    import socket
    import subprocess
    import torch
    n = args.gpus
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:
        env["DRFE_BENCH_ONE_DEVICE"] = "1"
        env["DRFE_BENCH_BACKEND"] = "gloo"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = procs[0].stdout.read().decode()
    rc = 0
    deadline = time.time() + 1800
    for p in procs:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
        rc = rc or p.returncode
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
        sys.stderr.write(out0)
        return rc or 1
    sys.stdout.write(out0)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per step per GPU")
    ap.add_argument("--config", type=int, default=0, choices=(0, 2, 4),
                    help="BASELINE.json config: 2 = TUM3 single sequence, 4 = TUM1/2/3 mix, one 256-frame sequence "
                         "per rank; 0 = config 2 at one rank, config 4 at N > 1")
    ap.add_argument("--distinct", type=int, default=0,
                    help="frames rendered per rank (0 = the config's sequence length: 64 / 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bow", action="store_true", help="also run the vocabulary tree descent in every step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is used\n")
    config = args.config or (2 if world == 1 else 4)

    # Render this rank's sequence first: the pool forks, so it must run before anything touches the GPU.
    from dr_slam_amd import sharding
    seed, cam, kind, seq_len = sharding.rank_workload(config, rank)
    n_distinct = args.distinct or seq_len
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    base = sharding.render_sequence(seed, n_distinct, cam, kind, workers=max(1, sharding.host_cpus() // local_world))

    import torch
    import torch.distributed as dist

    # Rehearsal hooks for a box with fewer devices than ranks (set by launch(); the driver never sets them):
    # DRFE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and DRFE_BENCH_BACKEND=gloo replaces RCCL, which refuses
    # two ranks on one device.
    rehearsal = bool(os.environ.get("DRFE_BENCH_ONE_DEVICE"))
    if rehearsal:
        local_rank = 0
    backend = os.environ.get("DRFE_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from dr_slam_amd.pipeline import FrontEnd

    B = args.batch
    gray, depth, Tcw, Twc = make_batch(base, B)
    fe = FrontEnd(cam, max_batch=B, device=local_rank)
    gray_t = torch.from_numpy(gray).to(dev)
    depth_t = torch.from_numpy(depth.view(np.int16)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    if world > 1 or args.bow:
        # the one initial exchange of the sharded mode: rank 0 owns the ORB vocabulary
        # (k=10, L=6, ~50 MB flattened; synthetic because the reference's ORBvoc blob is missing) and
        # broadcasts it over RCCL/xGMI; every rank uploads its copy into its own context.
        from dr_slam_amd import vocabulary as V
        if rank == 0:
            blob = V.make_synthetic(10, 6, seed=1).pack()
            size = np.array([blob.size], np.int64)
        else:
            size = np.zeros(1, np.int64)
        size = sharding.broadcast_tables(size, dev, dist if world > 1 else None)
        if rank != 0:
            blob = np.zeros(int(size[0]), np.uint8)
        blob = sharding.broadcast_tables(blob, dev, dist if world > 1 else None)
        V.Vocabulary.unpack(blob).upload(fe.ctx)

    def step():
        fe.process(gray_t, depth_t, Tcw, Twc, th=15.0, check_ori=True, stream=stream)
        if args.bow:   # Frame::ComputeBoW tree descent for every frame of the batch (not part of the metric)
            fe.ctx.bow_transform_batch(4, B, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    el, total_frames = sharding.reduce_elapsed_and_frames(el, B * args.steps, dev, dist if world > 1 else None)

    # sanity: the batch really produced keypoints and matches
    counts = fe.ctx.orb_counts(B)
    _, nm = fe.matches(B - 1)
    if not os.environ.get("DRFE_BENCH_NO_SANITY"):      # kernel experiments with deliberately wrong results
        assert counts.min() > 500 and nm > 100, (counts.min(), nm)

    out = None
    if rank == 0:
        # per-stage kernel time: same steps again with HIP events around every stage on the launch stream
        fe.ctx.profile_enable(True)
        acc = {}
        reps = max(3, min(args.steps, 10))
        for _ in range(reps):
            step()
            ms = fe.ctx.profile_stage_ms()
            for k, v in ms.items():
                acc[k] = acc.get(k, 0.0) + v
        fe.ctx.profile_enable(False)
        stage_ms = {k: v / reps for k, v in acc.items()}
        cand = {k: stage_ms[k] for k in ALGO_BYTES}
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        roof_stage = dom if dom in ALGO_BYTES else max(cand, key=lambda k: cand[k])
        algo = ALGO_BYTES[roof_stage] * B
        achieved = algo / (stage_ms[roof_stage] * 1e-3) / 1e9
        # HBM traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # separate runs; cannot be collected from inside this process). Only valid for the same batch size.
        traffic, traffic_src = None, None
        kname = {"pyramid": "k_pyr_resize", "fast": "k_fast_cells", "blur": "k_blur", "desc": "k_orient_desc",
                 "match": "k_window_candidates"}[roof_stage]
        try:
            pmc_name = "r01_pmc_traffic_b%d.json" % B
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            if pmc.get("batch") == B and kname in pmc["kernels"] and roof_stage != "pyramid":
                k = pmc["kernels"][kname]
                traffic = float(k["HBM_BYTES_per_launch"])
                traffic_src = "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE: the gfx950 half-count correction, calibrated, see its _about)" % pmc_name
        except Exception:
            pass
        fps = total_frames / el
        out = {
            "metric": "RGB-D frames/sec (extract+match) at 640x480",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": WORKLOAD_TEXT[config], "baseline_config": config, "batch_per_gpu": B,
                       "frames_per_step": world * B, "sequence_frames_per_rank": len(base),
                       "sharding": "one sequence per GPU, no data-path collective"},
            "stage_ms_per_batch": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": kname, "kernel_stage": roof_stage, "dominant_stage": dom,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": stage_ms[roof_stage],
                         "algorithmic_bytes_per_launch": algo},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is reported by the N=1 run only
            out["cpu_baseline"] = cpu_baseline(base, cam)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        if rehearsal:
            out["rehearsal"] = "%d ranks share ONE device over gloo: control-path check, not a scaling number" % world
        print(json.dumps(out))


if __name__ == "__main__":
    main()
