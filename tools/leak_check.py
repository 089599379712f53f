"""Creates a context, runs the three frame-batch entries on 64 frames, closes it - five times - and prints the device memory in use
after each round (a leak of an arena shows as a step)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dr_slam_amd import lib, synth
cam = synth.ICL
frames = list(synth.sequence(3, 8, cam=cam, kind="living_room"))
gray = np.stack([frames[i % 8][0] for i in range(64)]); depth = np.stack([frames[i % 8][1] for i in range(64)])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
depth_m = depth.astype(np.float32) * np.float32(inv)
torch.cuda.init()
for r in range(5):
    ctx = lib.Context(max_batch=1)
    ctx.lsd_extract_batch(gray, n_threads=4); ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=4); ctx.planes_cape_batch(depth_m, K4, 20, n_threads=2)
    free_in, total = torch.cuda.mem_get_info()
    ctx.close()
    free_out, _ = torch.cuda.mem_get_info()
    print("round %d: in use with the context %.2f GB, after close %.2f GB" % (r, (total - free_in) / 2**30, (total - free_out) / 2**30), flush=True)
