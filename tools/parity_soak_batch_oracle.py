#!/usr/bin/env python3
"""The frame-batch entries (sequential cores on the device) against the CPU ORACLE directly, frame by frame - key lines, LBD
descriptors, line equations; AHC planes, label images, post-processing; CAPE planes - over seeded frames of every scene kind.
Run on a GPU box: python tools/parity_soak_batch_oracle.py [frames per scene kind] [first seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PAIRS = [("angle", "angle"), ("response", "response"), ("start_point_x", "startPointX"), ("start_point_y", "startPointY"),
         ("end_point_x", "endPointX"), ("end_point_y", "endPointY"), ("line_length", "lineLength"), ("num_of_pixels", "numOfPixels")]


def main():
    from dr_slam_amd import lib, synth
    from oracle import oracle as O
    O.lib()
    per_kind = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    kinds = ["room_boxes", "planar_lowtexture", "living_room", "corridor"]
    cams = [synth.TUM3, synth.ICL]
    ctx = lib.Context(max_batch=1)
    bad = {"lines": 0, "planes": 0, "post": 0, "cape": 0}
    total = 0
    t0 = time.time()
    for ki, kind in enumerate(kinds):
        cam = cams[ki % 2]
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        inv = np.float32(1.0) / np.float32(cam.depth_factor)
        frames = []
        for s in range(0, per_kind, 8):
            frames += list(synth.sequence(seed0 + 17 * ki + s, min(8, per_kind - s), cam=cam, kind=kind, start=(s * 3) % 24))
        gray = np.stack([f[0] for f in frames]); depth = np.stack([f[1] for f in frames])
        depth_m = np.stack([O.depth_to_float(d, inv) for d in depth])
        lines = ctx.lsd_extract_batch(gray, n_threads=8)
        planes, n, post, na, pn, seg = ctx.planes_ahc_post_batch(depth, K4, float(inv), 9.0, 0.10, n_threads=8, seg=True)
        cplanes, cn, cseg = ctx.planes_cape_batch(depth_m, K4, 20, n_threads=4, seg=True)
        for f in range(len(frames)):
            a, b = lines[f], O.extract_lines(gray[f])
            ok = a["detected"] == b["detected"] and len(a["lines"]) == len(b["lines"]) and np.array_equal(a["desc"], b["desc"]) and \
                np.array_equal(a["lineF"].view(np.uint64), b["lineF"].view(np.uint64))
            for pa, pb in PAIRS:
                ok = ok and np.array_equal(a["lines"][pa].view(np.uint32), b["lines"][pb].view(np.uint32))
            if not ok:
                bad["lines"] += 1; print(f"MISMATCH lines: {kind} frame {f}", flush=True)
            oa = O.ahc_planes(depth[f], K4, float(inv))
            ok = n[f] == len(oa["planes"]) and np.array_equal(seg[f], oa["seg"]) and \
                np.array_equal(planes[f, :n[f]]["normal"].view(np.uint64), oa["planes"][:, 0:3].view(np.uint64)) and \
                np.array_equal(planes[f, :n[f]]["mse"].view(np.uint64), oa["planes"][:, 6].view(np.uint64))
            if not ok:
                bad["planes"] += 1; print(f"MISMATCH planes: {kind} frame {f}", flush=True)
            else:
                o2, opn = O.ahc_post_planes(depth[f], K4, float(inv), oa, 9.0, 0.10)
                ok = pn[f] == opn
                for k, rec in enumerate(o2):
                    ok = ok and bool(post[f, k]["accepted"]) == rec["accepted"] and post[f, k]["n_voxels"] == len(rec["voxels"]) and \
                        np.array_equal(post[f, k]["coef"].view(np.uint32), rec["coef"].view(np.uint32))
                if not ok:
                    bad["post"] += 1; print(f"MISMATCH post-processing: {kind} frame {f}", flush=True)
            oc = O.cape_planes(depth_m[f], K4, 20)
            if not (cn[f] == len(oc["planes"]) and np.array_equal(cseg[f], oc["seg"]) and
                    np.array_equal(cplanes[f, :cn[f]]["normal"].view(np.uint64), oc["planes"][:, 0:3].view(np.uint64))):
                bad["cape"] += 1; print(f"MISMATCH cape: {kind} frame {f}", flush=True)
        total += len(frames)
        print(f"{kind:18s} {len(frames)} frames against the oracle: {int(n.sum())} AHC planes, {int(cn.sum())} CAPE planes, {sum(len(x['lines']) for x in lines)} key lines", flush=True)
    print(f"{total} frames, mismatches {bad}, {time.time() - t0:.0f} s")
    print(f"frames the device handed back to the host: lines {ctx.lsd_stats()}, CAPE {ctx.planes_cape_stats()}, AHC extractor / voxel grids {ctx.planes_ahc_stats()}, "
          f"gates + refit {ctx.planes_refit_stats()}")
    ctx.close()
    return 1 if any(bad.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
