import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from dr_slam_amd import lib, synth
ctx = lib.Context(max_batch=1)
for ki, kind in enumerate(["room_boxes", "planar_lowtexture", "living_room", "corridor"]):
    cam = [synth.TUM3, synth.ICL][ki % 2]
    frames = []
    for s in range(0, 64, 8):
        frames += list(synth.sequence(7000 + 17 * ki + s, 8, cam=cam, kind=kind, start=(s * 3) % 24))
    gray = np.stack([f[0] for f in frames])
    s0 = ctx.lsd_stats()
    ctx.lsd_extract_batch(gray, n_threads=4)
    s1 = ctx.lsd_stats()
    print(kind, s1["nfa_to_host"] - s0["nfa_to_host"], "of", len(frames), flush=True)
