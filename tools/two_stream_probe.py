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
for B in (512, 256):
    gray, depth, Tcw, Twc = bench.make_batch(base, B)
    g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
    fes = [FrontEnd(cam, max_batch=B), FrontEnd(cam, max_batch=B)]
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    def run(n, two):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n):
            k = (i & 1) if two else 0
            fes[k].process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=ss[k].cuda_stream)
        torch.cuda.synchronize()
        return B * n / (time.perf_counter() - t)
    run(4, True)
    print("batch", B, "one stream", round(run(20, False)), "two streams", round(run(20, True)), "frames/s")
    for f in fes: f.ctx.close()
