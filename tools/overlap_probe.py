"""Does a 157 MB pinned H2D on one stream overlap with the ORB kernels on another on this box?  (experiment for the host-fed paths)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from dr_slam_amd import sharding, synth
from dr_slam_amd.pipeline import FrontEnd
cam = synth.TUM3
base = sharding.render_sequence(10, 8, cam, "room_boxes", workers=8)
B = 512
gray, depth, Tcw, Twc = bench.make_batch(base, B)
dev = torch.device("cuda", 0)
fe = FrontEnd(cam, max_batch=B)
gh = torch.from_numpy(gray).pin_memory()
gd = [torch.from_numpy(gray).to(dev) for _ in range(3)]
s_copy, s_comp = torch.cuda.Stream(), torch.cuda.Stream()
w, h = cam.w, cam.h
def run(n, copy, comp, dep):
    torch.cuda.synchronize(); t = time.perf_counter()
    evs = [torch.cuda.Event() for _ in range(n)]
    for i in range(n):
        if copy:
            with torch.cuda.stream(s_copy):
                gd[i % 3].copy_(gh, non_blocking=True)
                evs[i].record(s_copy)
        if comp:
            if dep and copy: s_comp.wait_event(evs[i])
            fe.ctx.orb_extract_batch_ptr(gd[(i + (0 if dep else 1)) % 3].data_ptr(), w * h, w, w, h, B, s_comp.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
run(3, True, True, False)
for name, a in (("copy only", (True, False, False)), ("compute only", (False, True, False)), ("both, independent", (True, True, False)), ("both, compute waits for its copy", (True, True, True))):
    print(name, round(run(10, *a), 2), "ms/step")
