"""Do two independent batches in flight (two contexts, two streams) raise the device rate?  (experiment)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from dr_slam_amd import sharding, synth
from dr_slam_amd.pipeline import FrontEnd
cam = synth.TUM3
base = sharding.render_sequence(10, 8, cam, "room_boxes", workers=8)
dev = torch.device("cuda", 0)
for B, S in ((512, 1), (384, 2), (256, 2), (192, 3), (128, 4), (512, 1), (384, 2), (256, 2), (192, 3), (128, 4)):
    gray, depth, Tcw, Twc = bench.make_batch(base, B)
    g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
    fes = [FrontEnd(cam, max_batch=B) for _ in range(S)]
    ss = [torch.cuda.Stream() for _ in range(S)]
    def run(n):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n):
            k = i % S
            fes[k].process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=ss[k].cuda_stream)
        torch.cuda.synchronize()
        return B * n / (time.perf_counter() - t)
    run(2 * S)
    print("batch", B, "x", S, "streams:", [round(run(12 * S)) for _ in range(3)], "frames/s")
    for f in fes: f.ctx.close()
