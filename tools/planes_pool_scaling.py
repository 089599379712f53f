"""drfe_planes_ahc_post_batch alone on 512 living-room frames against the number of host threads: thread-ms per frame =
wall * threads / frames; DRFE_TRACE_PLANES=1 adds the per-stage host times of one frame."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(512, len(base))
depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
ctx = lib.Context(max_batch=1)
for T in (1, 4, 8, 12, 16, 20):
    n = min(512, 64 * T)
    ctx.planes_ahc_post_batch(depth[:T * 2], K4, inv, 9.0, 0.10, n_threads=T)
    t = time.perf_counter(); c0 = time.process_time()
    ctx.planes_ahc_post_batch(depth[:n], K4, inv, 9.0, 0.10, n_threads=T)
    el = time.perf_counter() - t; cpu = time.process_time() - c0
    print("planes threads %2d: %3d frames, wall %7.1f ms, %5.2f thread-ms per frame, %5.2f CPU-ms per frame" % (T, n, el * 1e3, el * 1e3 * T / n, cpu * 1e3 / n), flush=True)
ctx.close()
