import sys, time, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from dr_slam_amd import sharding, synth
from dr_slam_amd.pipeline import FrontEnd
cam = synth.TUM3
base = sharding.render_sequence(10, 8, cam, "room_boxes", workers=8)
B = 512
gray, depth, Tcw, Twc = bench.make_batch(base, B)
dev = torch.device("cuda", 0)
fes = [FrontEnd(cam, max_batch=B), FrontEnd(cam, max_batch=B)]
K = fes[0].ctx.max_kp
# gather cost alone
g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
fes[0].process(g, d, Tcw, Twc, stream=torch.cuda.current_stream().cuda_stream)
uv = torch.empty((B, K), dtype=torch.int32).pin_memory(); kc = torch.empty(B, dtype=torch.int32).pin_memory()
fes[0].ctx.keypoint_pixels_async_ptr(B, uv.data_ptr(), kc.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
out = np.zeros((B, K), np.uint16)
for T in (1, 2, 4, 8):
    t = time.perf_counter()
    for _ in range(5): fes[0].ctx.gather_keypoint_depth(depth, uv.numpy().view(np.uint32), kc.numpy(), out, T)
    print("gather", T, "threads", (time.perf_counter() - t) / 5 * 1e3, "ms")
bench.host_fed_sparse_rate(fes, gray, depth, Tcw, Twc, B, 4, dev)
for thr in (1, 2, 4, 8):
    r = bench.host_fed_sparse_rate(fes, gray, depth, Tcw, Twc, B, 10, dev, _threads=thr)
    print("gather threads", thr, "->", round(r[0]), "frames/s", round(r[1], 2), "ms/step")
