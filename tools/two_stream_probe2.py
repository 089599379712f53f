"""Ramp of the two-batches-in-flight configuration: successive 20-step windows, contexts created before / after the inputs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from dr_slam_amd import sharding, synth
from dr_slam_amd.pipeline import FrontEnd
cam = synth.TUM3
base = sharding.render_sequence(10, 8, cam, "room_boxes", workers=8)
dev = torch.device("cuda", 0)
for B, S, order in ((512, 1, 'contexts first'), (512, 1, 'tensors first'), (384, 2, 'tensors first'), (512, 2, 'tensors first'), (256, 2, 'tensors first')):
  gray, depth, Tcw, Twc = bench.make_batch(base, B)
  if True:
    if order == "contexts first":
        fes = [FrontEnd(cam, max_batch=B) for _ in range(S)]
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
    else:
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
        fes = [FrontEnd(cam, max_batch=B) for _ in range(S)]
    ss = [torch.cuda.Stream() for _ in range(S)]
    n = [0]
    def run(k):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k):
            i = n[0] % S; n[0] += 1
            fes[i].process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=ss[i].cuda_stream)
        torch.cuda.synchronize()
        return round(B * k / (time.perf_counter() - t))
    run(3)
    print(B, S, order, [run(20) for _ in range(6)])
    for f in fes: f.ctx.close()
    del g, d, fes
