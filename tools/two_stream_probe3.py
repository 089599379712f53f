"""Reliability of two batches in flight: fresh processes, contexts allocated before / after the input tensors, 8 windows of 20 steps.
    python tools/two_stream_probe3.py            (driver: spawns the trials)
    python tools/two_stream_probe3.py ORDER B S  (one trial)"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())


def trial(order, B, S):
    import torch, bench
    from dr_slam_amd import sharding, synth
    from dr_slam_amd.pipeline import FrontEnd
    cam = synth.TUM3
    base = sharding.render_sequence(10, 8, cam, "room_boxes", workers=8)
    dev = torch.device("cuda", 0)
    gray, depth, Tcw, Twc = bench.make_batch(base, B)
    if order == "contexts":
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
    print(order, B, S, [run(20) for _ in range(8)], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        trial(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    else:
        for rep in range(3):
            for order in ("contexts", "tensors"):
                for B, S in ((512, 2), (384, 2)):
                    subprocess.run([sys.executable, __file__, order, str(B), str(S)], stderr=subprocess.DEVNULL)
