"""Per-frame host cost of the two thread-pooled batch entry points against the number of host threads (GPU box):
drfe_lsd_extract_batch and drfe_planes_ahc_post_batch on 96 living-room frames.  thread-ms per frame = wall * threads / frames:
flat = the pool scales, rising = the threads get in each other's way (memory bandwidth, allocator, device lanes)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(96, len(base))
gray = np.stack([base[i][0] for i in order]); depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
ctx = lib.Context(max_batch=1)
for T in (1, 2, 4, 8, 16, 20, 24):
    ctx.lsd_extract_batch(gray[:T * 2], n_threads=T)
    t = time.perf_counter(); ctx.lsd_extract_batch(gray, n_threads=T); el = time.perf_counter() - t
    print("lines  threads %2d: wall %7.1f ms, %5.2f thread-ms per frame" % (T, el * 1e3, el * 1e3 * T / len(gray)), flush=True)
for T in (1, 2, 4, 8, 16, 20, 24):
    ctx.planes_ahc_post_batch(depth[:T * 2], K4, inv, 9.0, 0.10, n_threads=T)
    t = time.perf_counter(); ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=T); el = time.perf_counter() - t
    print("planes threads %2d: wall %7.1f ms, %5.2f thread-ms per frame" % (T, el * 1e3, el * 1e3 * T / len(depth)), flush=True)
ctx.close()
