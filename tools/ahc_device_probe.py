"""drfe_planes_ahc_post_batch with the extractor on the device against the host pool: frames redone on the host (stderr,
DRFE_TRACE_PLANES=1), wall time and CPU time per frame.  python tools/ahc_device_probe.py [frames] [threads]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(B, len(base))
depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
ctx = lib.Context(max_batch=1)
res = {}
for dev in (True, False, True):
    ctx.planes_configure_extractor(dev)
    ctx.planes_ahc_post_batch(depth[:min(B, 16)], K4, inv, 9.0, 0.10, n_threads=T)
    t = time.perf_counter(); c0 = time.process_time()
    out = ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=T)
    el = time.perf_counter() - t; cpu = time.process_time() - c0
    res[dev] = out
    print("%s extractor: %d frames in %.1f ms = %.0f frames/s; CPU %.2f ms per frame; planes of frame 0: %d, accepted %d" %
          ("device" if dev else "host  ", B, el * 1e3, B / el, cpu * 1e3 / B, out[1][0], out[3][0]), flush=True)
same = all(np.array_equal(np.asarray(a).view(np.uint8) if hasattr(a, "view") else a, np.asarray(b).view(np.uint8) if hasattr(b, "view") else b) for a, b in zip(res[True], res[False]))
print("device == host:", same)
ctx.close()
