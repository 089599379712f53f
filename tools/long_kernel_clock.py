"""Kernel durations of the line / plane batch entries, one 512-frame call alone on the device (drfe_long_kernel_clock: HIP events on the
launch stream), then the path's saturation rate.   python tools/long_kernel_clock.py lines|planes [frames] [steps in flight for the rate]"""
import os, sys, threading, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
path = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
flights = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(N, len(base))
gray = np.stack([base[i][0] for i in order]); depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
T = 2 if path == "lines" else 6
def step(ctx):
    if path == "lines": return ctx.lsd_extract_batch(gray, n_threads=T)
    return ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=T)
c = lib.Context(max_batch=1)
step(c)
c.long_kernel_clock(True)
for _ in range(3):
    step(c)
    ms = c.long_kernel_ms()
    print(path, N, "frames alone:", {k: round(v, 2) for k, v in ms.items() if v > 0 and k.startswith("k_" if True else "")}, flush=True)
c.long_kernel_clock(False)
c.close()
if flights > 0:
    ctxs = [lib.Context(max_batch=1) for _ in range(flights)]
    for x in ctxs: step(x)
    reps = 3
    def run(x):
        for _ in range(reps): step(x)
    th = [threading.Thread(target=run, args=(x,)) for x in ctxs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print("%s: %d steps in flight: %6.0f frames/s" % (path, flights, flights * reps * N / el), flush=True)
    for x in ctxs: x.close()
