"""What makes two batches in flight fast (224 k) or not (206 k)?  Fresh process per variant.
    python tools/streams_in_flight_probe.py [VARIANT]"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())


def trial(variant, B=512, S=2):
    import torch, bench
    from dr_slam_amd import sharding, synth
    from dr_slam_amd.pipeline import FrontEnd
    cam = synth.TUM3
    base = sharding.render_sequence(10, 8, cam, "room_boxes", workers=8)
    dev = torch.device("cuda", 0)
    gray, depth, Tcw, Twc = bench.make_batch(base, B)
    keep = []
    if variant == "realloc":          # contexts created, closed, created again
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
        tmp = [FrontEnd(cam, max_batch=B) for _ in range(S)]
        for f in tmp: f.ctx.close()
        fes = [FrontEnd(cam, max_batch=B) for _ in range(S)]
    elif variant == "prior_run":      # what probe2 did: a complete single-context run first, everything freed, then the pair
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
        f0 = FrontEnd(cam, max_batch=B)
        for _ in range(10): f0.process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=0)
        torch.cuda.synchronize(); f0.ctx.close(); del g, d
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
        fes = [FrontEnd(cam, max_batch=B) for _ in range(S)]
    elif variant == "dummy4g":        # a 4 GB allocation in front of everything
        keep.append(torch.empty(4 << 30, dtype=torch.uint8, device=dev))
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
        fes = [FrontEnd(cam, max_batch=B) for _ in range(S)]
    elif variant == "interleave":     # context, padding, context
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
        fes = [FrontEnd(cam, max_batch=B)]
        keep.append(torch.empty(1 << 30, dtype=torch.uint8, device=dev))
        fes.append(FrontEnd(cam, max_batch=B))
    else:
        g = torch.from_numpy(gray).to(dev); d = torch.from_numpy(depth.view(np.int16)).to(dev)
        fes = [FrontEnd(cam, max_batch=B) for _ in range(S)]
    if variant == "ctx_streams":
        cs = [0, 0]
    elif variant.startswith("skip"):      # skipN: create N streams between the two that are used
        k = int(variant[4:])
        pool = [torch.cuda.Stream() for _ in range(k + 2)]
        cs = [pool[0].cuda_stream, pool[-1].cuda_stream]
        keep.append(pool)
    else:
        ss = [torch.cuda.Stream() for _ in range(S)]
        cs = [x.cuda_stream for x in ss]
        keep.append(ss)
    n = [0]

    def run(k):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k):
            i = n[0] % S; n[0] += 1
            fes[i].process(g, d, Tcw, Twc, th=15.0, check_ori=True, stream=cs[i])
        torch.cuda.synchronize()
        return round(B * k / (time.perf_counter() - t))
    run(3)
    print(variant, [run(20) for _ in range(6)], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        trial(sys.argv[1])
    else:
        for rep in range(2):
            for v in ("plain", "ctx_streams", "skip1", "skip2", "skip3", "skip5"):
                subprocess.run([sys.executable, __file__, v], stderr=subprocess.DEVNULL)
            for q in ("2", "8"):
                subprocess.run([sys.executable, __file__, "plain"], stderr=subprocess.DEVNULL, env=dict(os.environ, GPU_MAX_HW_QUEUES=q))
