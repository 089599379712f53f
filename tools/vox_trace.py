import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 4, cam, "living_room", workers=1)
depth = np.stack([b[1] for b in base])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
ctx = lib.Context(max_batch=1)
for rep in range(2):
    ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=1)
ctx.close()
