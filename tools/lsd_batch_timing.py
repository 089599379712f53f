"""Throughput of drfe_lsd_extract_batch with region growing on the device (k_lsd_grow) and on the host pool:
python tools/lsd_batch_timing.py [frames] [threads].  DRFE_TRACE_LINES=1 prints the batch's own phase times."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dr_slam_amd import lib, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 0
base = [f[0] for f in synth.sequence(3, 32, cam=synth.ICL, kind="living_room")]
gray = np.stack([base[i % len(base)] for i in range(B)])
ctx = lib.Context(max_batch=1)
for dev in (True, False, True):
    ctx.lsd_configure(dev)
    n = B if dev else min(B, 96)
    ctx.lsd_extract_batch(gray[:min(n, 32)], n_threads=T)
    t = time.perf_counter(); c0 = time.process_time()
    out = ctx.lsd_extract_batch(gray[:n], n_threads=T)
    el = time.perf_counter() - t; cpu = time.process_time() - c0
    print("%s grow: %d frames in %.1f ms = %.0f frames/s; host CPU %.2f ms per frame; lines of frame 0: %d" %
          ("device" if dev else "host  ", n, el * 1e3, n / el, cpu / n * 1e3, len(out[0]["lines"])), flush=True)
