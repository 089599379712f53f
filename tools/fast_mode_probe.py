"""Does a stage's time depend on the context, i.e. on where its arenas happen to lie?  (The FAST stage's did: 0.59 / 0.66 / 0.69 /
0.71 ms for the same batch until its atomic counters got a cache line each.)  Six contexts alive at once in ONE process, every
stage of each timed with HIP events."""
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


def stage_ms(fe):
    for _ in range(3):
        fe.process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=0)
    fe.ctx.profile_enable(True)
    acc = {}
    for _ in range(5):
        fe.process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=0)
        for k, v in fe.ctx.profile_stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 5
    fe.ctx.profile_enable(False)
    return acc


fes = [FrontEnd(cam, max_batch=B) for _ in range(6)]
stage_ms(fes[0])                                       # clocks up before the first measurement
res = [stage_ms(f) for f in fes]
print("six contexts alive at once, stage ms per context:")
for k in res[0]:
    print(f"  {k:10s}", [round(r[k], 3) for r in res])
