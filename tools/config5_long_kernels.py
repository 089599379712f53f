"""BASELINE config 5 (1280 x 960): one call of the line batch entry and one of the plane batch entry, 128 frames each - the command the
round-6 kernel statistics of k_lsd_order / k_lsd_grow / k_ahc_cluster_big / k_ahc_refine_big at that size are collected on
(rocprofv3 --kernel-trace --stats -- python3 tools/config5_long_kernels.py), and the same durations by the library's own clock."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
seed, cam, kind, seq_len = sharding.rank_workload(5, 0)
base = sharding.render_sequence(seed, 16, cam, kind, workers=1)
gray = np.stack([base[i % len(base)][0] for i in range(N)]); depth = np.stack([base[i % len(base)][1] for i in range(N)])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
c = lib.Context(max_width=cam.w, max_height=cam.h, max_batch=1)
c.lsd_extract_batch(gray, n_threads=2); c.planes_ahc_post_batch(depth, K4, inv, 5.0, 0.10)      # arenas
c.long_kernel_clock(True)
for _ in range(2):
    t0 = time.perf_counter(); c.lsd_extract_batch(gray, n_threads=2); tl = time.perf_counter() - t0
    t0 = time.perf_counter(); c.planes_ahc_post_batch(depth, K4, inv, 5.0, 0.10); tp = time.perf_counter() - t0
    print("%d x %d, %d frames per call: lines %.1f ms (%.0f frames/s), planes %.1f ms (%.0f frames/s); kernels by the library's clock (ms): %s" %
          (cam.w, cam.h, N, tl * 1e3, N / tl, tp * 1e3, N / tp, {k: round(v, 2) for k, v in c.long_kernel_ms().items() if v > 0}), flush=True)
print("handed back to the host:", c.lsd_stats(), c.planes_ahc_stats(), c.planes_refit_stats())
c.close()
