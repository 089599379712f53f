import sys, time, numpy as np
sys.path.insert(0, '.')
from dr_slam_amd import lib, synth
from oracle import oracle as O
cam = synth.TUM3
_, d, _ = next(synth.sequence(2, 1, kind="room_boxes"))
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
o = O.ahc_planes(d, K4, inv)
lib.planes_ahc_from_blocks(o["blocks"], o["block_valid"], o["block_N"], d, K4, inv)
t = time.perf_counter()
for _ in range(10): g = lib.planes_ahc_from_blocks(o["blocks"], o["block_valid"], o["block_N"], d, K4, inv)
print((time.perf_counter() - t) / 10 * 1e3, "ms", len(g["planes"]), "planes; valid blocks", int(o["block_valid"].sum()))
