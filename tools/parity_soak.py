#!/usr/bin/env python3
"""Parity soak: many seeded synthetic sequences of every scene kind through the batched product path (ORB extract + stereo /
grid + SearchByProjection against the previous frame) and through the CPU oracle, frame by frame; prints the number of
frames compared and every mismatch.  Longer than the test-suite cases; run on a GPU box:
    python tools/parity_soak.py [n_seq] [--cam TUM3|TUM1|TUM2|ICL|REALSENSE[xSCALE]] [--flow batch|frame|pipe]
--cam REALSENSEx2 is BASELINE config 5's 1280x960 stream, TUM1 / TUM2 have lens distortion (mvKeysUn live);
--flow frame pushes the frames one at a time through drfe_frame_submit / drfe_frame_collect on a two-slot context (frame k + 1
submitted once frame k has been matched against frame k - 1, which must stay in its slot until then) and matches slot pairs with
drfe_search_by_projection_last; --flow pipe keeps three batches in flight through drfe_pipeline_submit (three contexts round robin,
each on its own stream) and checks each batch when its context is about to be reused."""
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
    args = [a for a in sys.argv[1:]]
    camname, flow = "TUM3", "batch"
    if "--cam" in args:
        camname = args[args.index("--cam") + 1]
        del args[args.index("--cam"):args.index("--cam") + 2]
    if "--flow" in args:
        flow = args[args.index("--flow") + 1]
        del args[args.index("--flow"):args.index("--flow") + 2]
    n_seq = int(args[0]) if args else 12
    frames_per = 8
    base, _, sc = camname.partition("x")
    cam = getattr(synth, base)
    if sc:
        cam = cam.scaled(float(sc))
    dist = cam.dist if (len(cam.dist) > 0 and cam.dist[0] != 0.0) else None
    from dr_slam_amd import lib
    pipe = None
    if flow == "pipe":
        pipe = lib.Pipeline(3, max_width=cam.w, max_height=cam.h, max_batch=frames_per)
        views = [FrontEnd(cam, max_batch=frames_per, ctx=c) for c in pipe.contexts]
        fe = views[0]
    else:
        fe = FrontEnd(cam, max_batch=frames_per if flow == "batch" else 2)
    o = O.OrbOracle()
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    kinds = ["room_boxes", "planar_lowtexture", "living_room", "corridor"]
    bad = 0
    total = 0
    t0 = time.time()
    for s in range(n_seq):
        kind = kinds[s % len(kinds)]
        frames = list(synth.sequence(1000 + s, frames_per, cam=cam, kind=kind, start=(s * 7) % 40))
        Twc = np.stack([f[2] for f in frames]).astype(np.float64)
        Tcw = np.linalg.inv(Twc).astype(np.float32)
        Twc = Twc.astype(np.float32)
        if flow == "pipe":
            if s % 3 == 0:                 # submit this sequence and the next two back to back: three batches in flight
                group = []
                for s2 in range(s, min(s + 3, n_seq)):
                    fr2 = list(synth.sequence(1000 + s2, frames_per, cam=cam, kind=kinds[s2 % len(kinds)], start=(s2 * 7) % 40))
                    Twc2 = np.stack([f[2] for f in fr2]).astype(np.float64)
                    Tcw2 = np.linalg.inv(Twc2).astype(np.float32)
                    g2 = torch.from_numpy(np.stack([f[0] for f in fr2])).cuda()
                    d2 = torch.from_numpy(np.stack([f[1] for f in fr2]).view(np.int16)).cuda()
                    group.append((g2, d2, Tcw2, Twc2.astype(np.float32)))
                torch.cuda.synchronize()
                ks = [pipe.submit(g2.data_ptr(), d2.data_ptr(), cam.w * cam.h, cam.w, cam.w, cam.h, T1, T2, fe.cam, 15.0, False, True, frames_per)
                      for g2, d2, T1, T2 in group]
            fe = views[ks[s % 3]]
        if flow == "batch":
            gray = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
            depth = torch.from_numpy(np.stack([f[1] for f in frames]).view(np.int16)).cuda()
            fe.process(gray, depth, Tcw, Twc, th=15.0, check_ori=True, stream=torch.cuda.current_stream().cuda_stream)
        of = []
        frame_matches = {}
        for i, (g, d, _) in enumerate(frames):
            kps, desc = o(g)
            if flow in ("batch", "pipe"):
                gk, gd = fe.keypoints(i)
            else:
                if i == 0:
                    fe.ctx.frame_submit(0, g, d, fe.cam)
                gk, gd = fe.ctx.frame_collect(i % 2)
            ok = len(gk) == len(kps) and np.array_equal(gd, desc) and gk.tobytes() == kps.tobytes()   # same record layout
            if not ok:
                bad += 1
                print(f"MISMATCH extract: seq {s} ({kind}) frame {i}: {len(gk)} vs {len(kps)} keypoints")
            of.append(O.FrameOracle(kps, desc, O.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor)), K4, cam.bf, cam.w, cam.h, o.scale,
                                    dist=dist))
            total += 1
            if flow == "frame":
                if i >= 1 and ok:       # match against LastFrame, still resident in the other slot, before that slot is reused
                    world, valid = of[i - 1].unproject(Twc[i - 1])
                    gmp = np.zeros(of[i - 1].N, lib.MAPPOINT_DTYPE)
                    gmp["valid"], gmp["obs_positive"], gmp["world"], gmp["desc"] = valid, 1, world, of[i - 1].desc
                    n_g, m_g = fe.ctx.search_by_projection_last(i % 2, (i - 1) % 2, Tcw[i], Tcw[i - 1], fe.cam, gmp, of[i].N, 15.0, False, True)
                    frame_matches[i] = (m_g, n_g)
                if i + 1 < frames_per:
                    fe.ctx.frame_submit((i + 1) % 2, frames[i + 1][0], frames[i + 1][1], fe.cam)
        for i in range(1, frames_per):
            world, valid = of[i - 1].unproject(Twc[i - 1])
            mp = np.zeros(of[i - 1].N, O.MAPPOINT_DTYPE)
            mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, of[i - 1].desc
            n_o, m_o = O.search_by_projection_last(of[i], of[i - 1], Tcw[i], Tcw[i - 1], mp, 15.0, False, True)
            if flow in ("batch", "pipe"):
                m_g, n_g = fe.matches(i)
            elif i in frame_matches:
                m_g, n_g = frame_matches[i]
            else:
                continue
            if n_g != n_o or not np.array_equal(m_g[:of[i].N], m_o):
                bad += 1
                print(f"MISMATCH match: seq {s} ({kind}) frame {i}: {n_g} vs {n_o}")
    print(f"{camname} {cam.w}x{cam.h}, {flow} flow: {total} frames, {n_seq * (frames_per - 1)} frame pairs compared in {time.time() - t0:.0f} s: {bad} mismatches")
    if pipe is not None:
        pipe.close()
    else:
        fe.ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
