"""drfe_planes_ahc_post_batch on 512 living-room frames: voxel grids on the host pool against behind the device extractor.
Wall time and CPU time per frame, for a few pool sizes; DRFE_TRACE_PLANES=1 adds the pipeline's own counts."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(512, len(base))
depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
ctx = lib.Context(max_batch=1)
ref = None
for mode in (0, 1):
    ctx.planes_configure(device_voxel_grid=mode)
    for T in (4, 8, 16):
        ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=T)
        t = time.perf_counter(); c0 = time.process_time()
        r = ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=T)
        el = time.perf_counter() - t; cpu = time.process_time() - c0
        if ref is None:
            ref = r[2].tobytes()
        assert r[2].tobytes() == ref
        print("voxel grids on the %s, %2d threads: 512 frames, wall %7.1f ms (%6.0f frames/s), %5.2f CPU-ms per frame" %
              ("device" if mode else "host", T, el * 1e3, 512 / el, cpu * 1e3 / 512), flush=True)
ctx.close()
