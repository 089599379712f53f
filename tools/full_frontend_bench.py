#!/usr/bin/env python3
"""Secondary measurement (not the headline metric of bench.py): the FULL per-frame front-end of BASELINE config 2 —
ORB extract + glue + SearchByProjection (batched, device-resident) plus LSD+LBD lines, CAPE planes and AHC planes for
every frame — on one MI355X.  The sequential host stages of the line / plane paths (region growing, NFA, clustering,
flood fill) run on a pool of host threads, one drfe context per thread, the way the reference runs its four
extractors on four threads (src/Frame.cc:116-126) — here across frames instead of across extractors.
Prints one JSON line.  Usage: python tools/full_frontend_bench.py [--frames 64] [--threads N]"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--line-threads", type=int, default=12)
    ap.add_argument("--plane-threads", type=int, default=4)
    a = ap.parse_args()
    import torch
    from dr_slam_amd import lib, synth
    from dr_slam_amd.pipeline import FrontEnd
    from dr_slam_amd.sharding import pingpong_order
    cam = synth.ICL
    ncores = os.cpu_count() or 1
    T = a.threads or 2
    base = list(synth.sequence(3, 8, cam=cam, kind="living_room"))
    order = pingpong_order(a.frames, 8)
    gray = np.stack([base[i][0] for i in order])
    depth = np.stack([base[i][1] for i in order])
    Twc = np.stack([base[i][2] for i in order]).astype(np.float64)
    Tcw = np.linalg.inv(Twc).astype(np.float32)
    Twc = Twc.astype(np.float32)
    fe = FrontEnd(cam, max_batch=a.frames)
    gray_t = torch.from_numpy(gray).cuda()
    depth_t = torch.from_numpy(depth.view(np.int16)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    depth_m = [(d.astype(np.float32) * np.float32(inv)) for d in depth]
    ctxs = [lib.Context(max_batch=1) for _ in range(T)]      # CAPE lanes (one context per Python thread)
    ctx_planes = lib.Context(max_batch=1)                    # owns the AHC thread pool

    def aux(k):
        c = ctxs[k]
        n = 0
        for i in range(k, a.frames, T):
            c.planes_cape(depth_m[i], K4, 20)
            n += 1
        return n

    def step(pool):
        futs = [pool.submit(aux, k) for k in range(T)]
        futs_l = pool.submit(lambda: fe.ctx.lsd_extract_batch(gray, n_threads=a.line_threads))
        futs_p = pool.submit(lambda: ctx_planes.planes_ahc_batch(depth, K4, inv, n_threads=a.plane_threads, members=False))
        fe.process(gray_t, depth_t, Tcw, Twc, th=15.0, check_ori=True, stream=stream)
        torch.cuda.synchronize()
        assert len(futs_l.result()) == a.frames and len(futs_p.result()) == a.frames
        return sum(f.result() for f in futs)

    with ThreadPoolExecutor(T + 2) as pool:
        step(pool)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            assert step(pool) == a.frames
        el = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"metric": "full front-end frames/s (ORB+match batched on device; LSD+LBD, CAPE, AHC per frame)",
                      "value": a.frames / el, "unit": "frames/s", "frames_per_step": a.frames, "ms_per_step": el * 1e3,
                      "host_threads": T, "line_threads": a.line_threads, "plane_threads": a.plane_threads, "host_cores": ncores, "camera": "ICL", "scene": "living_room"}))
    for c in ctxs + [ctx_planes]:
        c.close()
    fe.ctx.close()


if __name__ == "__main__":
    main()
