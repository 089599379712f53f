"""How far one path of the front-end scales with steps in flight (each on its own context and host pool): 512 living-room
frames per step.   python tools/path_saturation.py lines|planes|cape  [threads per step] [steps in flight ...]"""
import os, sys, threading, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
path = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flights = [int(a) for a in sys.argv[3:]] or [1, 2, 3]
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(512, len(base))
gray = np.stack([base[i][0] for i in order]); depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
depth_m = depth.astype(np.float32) * np.float32(inv)
def step(ctx):
    if path == "lines": ctx.lsd_extract_batch(gray, n_threads=T)
    elif path == "planes": ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=T)
    else: ctx.planes_cape_batch(depth_m, K4, 20, n_threads=T)
for n in flights:
    ctxs = [lib.Context(max_batch=1) for _ in range(n)]
    for c in ctxs: step(c)
    reps = 3
    def run(c):
        for _ in range(reps): step(c)
    th = [threading.Thread(target=run, args=(c,)) for c in ctxs]
    c0 = time.process_time(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0; cpu = time.process_time() - c0
    tot = n * reps * 512
    print("%s: %d steps in flight x %d threads: %6.0f frames/s, %.2f CPU-ms per frame" % (path, n, T, tot / el, cpu * 1e3 / tot), flush=True)
    for c in ctxs: c.close()
