"""A/B of the region growing kernels on the lines path: one 512-frame step at a time and three in flight, with the counters that
show whether any frame went back to the host.   python tools/lines_mw_ab.py [frames]"""
import os, sys, threading, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(B, len(base))
gray = np.stack([base[i][0] for i in order])
for n in (1, 3):
    ctxs = [lib.Context(max_batch=1) for _ in range(n)]
    for c in ctxs: c.lsd_extract_batch(gray, n_threads=8)
    reps = 3
    def run(c):
        for _ in range(reps): c.lsd_extract_batch(gray, n_threads=8)
    th = [threading.Thread(target=run, args=(c,)) for c in ctxs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print("waves=%s: %d steps in flight: %6.0f frames/s; stats %s" % (os.environ.get("DRFE_LSD_GROW_WAVES", "auto"), n, n * reps * B / el, ctxs[0].lsd_stats()), flush=True)
    for c in ctxs: c.close()
