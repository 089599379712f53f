"""Do the long one-wavefront-per-frame kernels of several contexts really run side by side?  n contexts x 64 frames of the line
batch (64 wavefronts of k_lsd_grow each: the device is nearly empty), started together from n threads: wall time against one
context alone.  If the streams behind them share hardware queues, the walls add.   python tools/queue_probe.py [frames=64]"""
import os, sys, threading, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(F, len(base))
gray = np.stack([base[i][0] for i in order]); depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
for what in ("lines", "planes", "mixed"):
    for n in (1, 2, 4, 8, 12):
        ctxs = [lib.Context(max_batch=1) for _ in range(n)]
        def run(k, c):
            if what == "lines" or (what == "mixed" and k % 2 == 0): c.lsd_extract_batch(gray, n_threads=2)
            else: c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=2)
        for k, c in enumerate(ctxs): run(k, c)
        best = 1e9
        for _ in range(3):
            th = [threading.Thread(target=run, args=(k, c)) for k, c in enumerate(ctxs)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            best = min(best, time.perf_counter() - t0)
        print("%-6s %2d contexts x %d frames side by side: %7.1f ms" % (what, n, F, best * 1e3), flush=True)
        for c in ctxs: c.close()
