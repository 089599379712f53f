"""Where the fixed cost of a short timed region goes: K steps over `nfl` contexts, then the time at which every context's stream drains.
python tools/tail_timeline.py [K] [nfl]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from dr_slam_amd import lib, sharding
from dr_slam_amd.pipeline import FrontEnd
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nfl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seed, cam, kind, seq_len = sharding.rank_workload(2, 0)
base = sharding.render_sequence(seed, seq_len, cam, kind, workers=8)
B = 512
pipe = lib.Pipeline(nfl, 1000, 1.2, 8, 20, 7, cam.w, cam.h, B, 0)
fes = [FrontEnd(cam, max_batch=B, device=0, ctx=c) for c in pipe.contexts]
inputs = [bench.make_batch(base, B, k * 11) for k in range(nfl)]
g = [torch.from_numpy(a[0]).cuda() for a in inputs]; d = [torch.from_numpy(a[1].view(np.int16)).cuda() for a in inputs]
torch.cuda.synchronize()
nxt = [0]
def step():
    j = nxt[0]
    pipe.submit(g[j].data_ptr(), d[j].data_ptr(), cam.w * cam.h, cam.w, cam.w, cam.h, inputs[j][2], inputs[j][3], fes[0].cam, 15.0, False, True, B)
    nxt[0] = (j + 1) % nfl
for _ in range(300): step()
torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize()
    first = nxt[0]
    t0 = time.perf_counter()
    for _ in range(K): step()
    t_sub = time.perf_counter() - t0
    done = {}
    # contexts in the order they should drain: the one that received the fewest batches first
    order = sorted(range(nfl), key=lambda c: -((c - first) % nfl))
    for c in order:
        pipe.sync(c); done[c] = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) * 1e3
    nb = {c: len([i for i in range(K) if (first + i) % nfl == c]) for c in range(nfl)}
    print("K=%d nfl=%d: all submitted after %.2f ms; contexts (batches: drained at ms): %s; total %.2f ms = %.3f ms/step" %
          (K, nfl, t_sub * 1e3, {c: (nb[c], round(done[c], 2)) for c in order}, tot, tot / K), flush=True)
