"""One context, one drfe_lsd_extract_batch call of N frames (N wavefronts of k_lsd_grow in one launch): frames/s against N.
    python tools/lines_big_batch.py 512 1024 2048"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
ctx = lib.Context(max_batch=1)
for N in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    order = sharding.pingpong_order(N, len(base))
    gray = np.stack([base[i][0] for i in order])
    ctx.lsd_extract_batch(gray, n_threads=2)
    t0 = time.perf_counter(); ctx.lsd_extract_batch(gray, n_threads=2); el = time.perf_counter() - t0
    print("lines, one call of %4d frames: %7.1f ms = %6.0f frames/s" % (N, el * 1e3, N / el), flush=True)
