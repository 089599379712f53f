"""The FAST stage's time depended on the context (0.59 / 0.66 / 0.69 / 0.71 ms for the same batch): contexts alive at once in ONE
process, the FAST stage of each timed with HIP events."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from dr_slam_amd import sharding, synth
from dr_slam_amd.pipeline import FrontEnd
cam = synth.TUM3
base = sharding.render_sequence(10, 8, cam, "room_boxes", workers=8)
dev = torch.device("cuda", 0)
B = 512
gray, depth, Tcw, Twc = bench.make_batch(base, B)
g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
torch.cuda.synchronize()


def fast_ms(fe):
    for _ in range(3):
        fe.process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=0)
    fe.ctx.profile_enable(True)
    acc = []
    for _ in range(5):
        fe.process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=0)
        acc.append(fe.ctx.profile_stage_ms()["fast"])
    fe.ctx.profile_enable(False)
    return round(float(np.mean(acc)), 3)


fes = [FrontEnd(cam, max_batch=B) for _ in range(6)]
print("six contexts alive at once, FAST stage ms:", [fast_ms(f) for f in fes])
