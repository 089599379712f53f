#!/usr/bin/env python3
"""Parity soak of the frame-batch entries whose sequential cores run on the device (drfe_lsd_extract_batch,
drfe_planes_ahc_post_batch, drfe_planes_cape_batch) against the single-frame entries - the path tools/parity_soak_aux.py and the
-m gpu tests hold to the CPU oracle - over many seeded frames of every scene kind, and how many frames the device handed back
to the host.  Run on a GPU box: python tools/parity_soak_batch.py [frames per scene kind] [first seed] [image scale]   (DRFE_TRACE_LINES / _PLANES=1: counts)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from dr_slam_amd import lib, synth
    per_kind = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000          # other seeds: other sequences
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0           # camera / image scale (0.5: 320 x 240)
    kinds = ["room_boxes", "planar_lowtexture", "living_room", "corridor"]
    cams = [synth.TUM3, synth.ICL]
    ctx = lib.Context(max_batch=1)
    bad = {"lines": 0, "planes": 0, "post": 0, "cape": 0}
    total = 0
    t0 = time.time()
    for ki, kind in enumerate(kinds):
        cam = cams[ki % 2] if scale == 1.0 else cams[ki % 2].scaled(scale)
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
        frames = []
        for s in range(0, per_kind, 8):                       # eight consecutive frames of several seeded sequences
            frames += list(synth.sequence(seed0 + 17 * ki + s, min(8, per_kind - s), cam=cam, kind=kind, start=(s * 3) % 24))
            if s % 64 == 56:
                print(f"  {kind}: {len(frames)} frames rendered", flush=True)
        gray = np.stack([f[0] for f in frames]); depth = np.stack([f[1] for f in frames])
        depth_m = depth.astype(np.float32) * np.float32(inv)
        B = len(frames)
        total += B
        lb = ctx.lsd_extract_batch(gray, n_threads=8)
        planes, n, post, na, pn, seg = ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=8, seg=True)
        cplanes, cn, cseg = ctx.planes_cape_batch(depth_m, K4, 20, n_threads=4, seg=True)
        for f in range(B):
            if f % 64 == 63:
                print(f"  {kind}: {f + 1} frames compared", flush=True)
            a = ctx.lsd_extract(gray[f])
            ok = a["detected"] == lb[f]["detected"] and a["lines"].tobytes() == lb[f]["lines"].tobytes() and \
                np.array_equal(a["desc"], lb[f]["desc"]) and a["lineF"].tobytes() == lb[f]["lineF"].tobytes()
            if not ok:
                bad["lines"] += 1
                print(f"MISMATCH lines: {kind} frame {f}")
            ga = ctx.planes_ahc(depth[f], K4, inv)
            ok = n[f] == len(ga["planes"]) and planes[f, :n[f]].tobytes() == ga["planes"].tobytes() and np.array_equal(seg[f], ga["seg"])
            if not ok:
                bad["planes"] += 1
                print(f"MISMATCH planes: {kind} frame {f}")
            else:
                g = ctx.planes_ahc_postprocess(depth[f], K4, inv, ga, 9.0, 0.10)
                if not (post[f, :n[f]].tobytes() == g["post"].tobytes() and na[f] == g["n_accepted"] and pn[f] == g["plane_num"]):
                    bad["post"] += 1
                    print(f"MISMATCH post-processing: {kind} frame {f}")
            gc = ctx.planes_cape(depth_m[f], K4, 20)
            if not (cn[f] == len(gc["planes"]) and cplanes[f, :cn[f]].tobytes() == gc["planes"].tobytes() and np.array_equal(cseg[f], gc["seg"])):
                bad["cape"] += 1
                print(f"MISMATCH cape: {kind} frame {f}")
        print(f"{kind:18s} {cam.name if hasattr(cam, 'name') else ''} {B} frames: lines {sum(len(x['lines']) for x in lb)} key lines, "
              f"{int(n.sum())} AHC planes ({int(na.sum())} accepted), {int(cn.sum())} CAPE planes", flush=True)
    print(f"{total} frames, mismatches {bad}, {time.time() - t0:.0f} s")
    ctx.close()
    return 1 if any(bad.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
