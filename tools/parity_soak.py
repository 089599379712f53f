#!/usr/bin/env python3
"""Parity soak: many seeded synthetic sequences of every scene kind through the batched product path (ORB extract + stereo /
grid + SearchByProjection against the previous frame) and through the CPU oracle, frame by frame; prints the number of
frames compared and every mismatch.  Longer than the test-suite cases; run on a GPU box: python tools/parity_soak.py [n_seq]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from dr_slam_amd import synth
    from dr_slam_amd.pipeline import FrontEnd
    from oracle import oracle as O
    O.lib()
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    frames_per = 8
    cam = synth.TUM3
    fe = FrontEnd(cam, max_batch=frames_per)
    o = O.OrbOracle()
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    kinds = ["room_boxes", "planar_lowtexture", "living_room", "corridor"]
    bad = 0
    total = 0
    t0 = time.time()
    for s in range(n_seq):
        kind = kinds[s % len(kinds)]
        frames = list(synth.sequence(1000 + s, frames_per, kind=kind, start=(s * 7) % 40))
        gray = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
        depth = torch.from_numpy(np.stack([f[1] for f in frames]).view(np.int16)).cuda()
        Twc = np.stack([f[2] for f in frames]).astype(np.float64)
        Tcw = np.linalg.inv(Twc).astype(np.float32)
        Twc = Twc.astype(np.float32)
        fe.process(gray, depth, Tcw, Twc, th=15.0, check_ori=True, stream=torch.cuda.current_stream().cuda_stream)
        of = []
        for i, (g, d, _) in enumerate(frames):
            kps, desc = o(g)
            gk, gd = fe.keypoints(i)
            ok = len(gk) == len(kps) and np.array_equal(gd, desc) and gk.tobytes() == kps.tobytes()   # same record layout
            if not ok:
                bad += 1
                print(f"MISMATCH extract: seq {s} ({kind}) frame {i}: {len(gk)} vs {len(kps)} keypoints")
            of.append(O.FrameOracle(kps, desc, O.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor)), K4, cam.bf, cam.w, cam.h, o.scale))
            total += 1
        for i in range(1, frames_per):
            world, valid = of[i - 1].unproject(Twc[i - 1])
            mp = np.zeros(of[i - 1].N, O.MAPPOINT_DTYPE)
            mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, of[i - 1].desc
            n_o, m_o = O.search_by_projection_last(of[i], of[i - 1], Tcw[i], Tcw[i - 1], mp, 15.0, False, True)
            m_g, n_g = fe.matches(i)
            if n_g != n_o or not np.array_equal(m_g[:of[i].N], m_o):
                bad += 1
                print(f"MISMATCH match: seq {s} ({kind}) frame {i}: {n_g} vs {n_o}")
    print(f"{total} frames, {n_seq * (frames_per - 1)} frame pairs compared in {time.time() - t0:.0f} s: {bad} mismatches")
    fe.ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
