#!/usr/bin/env python3
"""Parity soak of the single-frame line and plane paths: LSD+LBD key lines / descriptors / line equations, the AHC plane
list / label image with its post-processing (voxel clouds, gates, refit coefficients), the CAPE planes / label image and the
surface normals - product (device kernels + host stages, through the C-ABI) against the CPU oracle, over many seeded frames
of every scene kind.  Run on a GPU box: python tools/parity_soak_aux.py [n_frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PAIRS = [("angle", "angle"), ("class_id", "class_id"), ("octave", "octave"), ("pt_x", "ptX"), ("pt_y", "ptY"),
         ("response", "response"), ("size", "size"), ("start_point_x", "startPointX"), ("start_point_y", "startPointY"),
         ("end_point_x", "endPointX"), ("end_point_y", "endPointY"), ("s_point_in_octave_x", "sPointInOctaveX"),
         ("s_point_in_octave_y", "sPointInOctaveY"), ("e_point_in_octave_x", "ePointInOctaveX"),
         ("e_point_in_octave_y", "ePointInOctaveY"), ("line_length", "lineLength"), ("num_of_pixels", "numOfPixels")]


def main():
    from dr_slam_amd import lib, synth
    from oracle import oracle as O
    O.lib()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    kinds = ["room_boxes", "planar_lowtexture", "living_room", "corridor"]
    ctx = lib.Context(max_batch=1)
    cam = synth.TUM3
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = np.float32(1.0) / np.float32(cam.depth_factor)
    bad = 0
    t0 = time.time()
    for i in range(n):
        kind = kinds[i % 4]
        g, d, _ = next(synth.sequence(2000 + i, 1, kind=kind, start=(i * 5) % 30))
        a, o = ctx.lsd_extract(g), O.extract_lines(g)
        ok = a["detected"] == o["detected"] and len(a["lines"]) == len(o["lines"]) and np.array_equal(a["desc"], o["desc"]) and \
            np.array_equal(a["lineF"].view(np.uint64), o["lineF"].view(np.uint64))
        for gk, okk in PAIRS:
            ok = ok and np.array_equal(a["lines"][gk].view(np.uint32), o["lines"][okk].view(np.uint32))
        if not ok:
            bad += 1
            print(f"MISMATCH lines: frame {i} ({kind})")
        gp, op = ctx.planes_ahc(d, K4, inv), O.ahc_planes(d, K4, inv)
        ok = len(gp["planes"]) == len(op["planes"]) and np.array_equal(gp["seg"], op["seg"]) and np.array_equal(gp["planes"]["n_points"], op["N"])
        if ok:
            ok = np.array_equal(gp["planes"]["normal"].view(np.uint64), op["planes"][:, 0:3].view(np.uint64)) and \
                np.array_equal(gp["planes"]["center"].view(np.uint64), op["planes"][:, 3:6].view(np.uint64)) and \
                np.array_equal(gp["planes"]["mse"].view(np.uint64), op["planes"][:, 6].view(np.uint64))
        if not ok:
            bad += 1
            print(f"MISMATCH planes: frame {i} ({kind})")
        else:
            for maxd, th in ((9.0, 0.10), (9.0, 0.05)):
                g2, (o2, pn) = ctx.planes_ahc_postprocess(d, K4, inv, gp, maxd, th), O.ahc_post_planes(d, K4, inv, op, maxd, th)
                ok = g2["plane_num"] == pn
                for k, rec in enumerate(o2):
                    ok = ok and bool(g2["post"]["accepted"][k]) == rec["accepted"] and g2["post"]["n_voxels"][k] == len(rec["voxels"]) and \
                        np.array_equal(g2["post"]["coef"][k].view(np.uint32), rec["coef"].view(np.uint32)) and \
                        (not rec["accepted"] or np.array_equal(g2["voxels"][k].view(np.uint32), rec["voxels"].view(np.uint32)))
                if not ok:
                    bad += 1
                    print(f"MISMATCH plane post-processing: frame {i} ({kind}) th {th}")
        dm = O.depth_to_float(d, inv)
        gc, oc = ctx.planes_cape(dm, K4, 20), O.cape_planes(dm, K4, 20)
        ok = len(gc["planes"]) == len(oc["planes"]) and np.array_equal(gc["seg"], oc["seg"]) and \
            np.array_equal(gc["planes"]["normal"].view(np.uint64), oc["planes"][:, 0:3].view(np.uint64)) and \
            np.array_equal(gc["planes"]["d"].view(np.uint64), oc["planes"][:, 6].view(np.uint64))
        if not ok:
            bad += 1
            print(f"MISMATCH CAPE planes: frame {i} ({kind})")
        rec = ctx.surface_normals(dm, K4, 9.0)
        ocl, onr = O.post_surface_normals(dm, K4, 9.0)
        on, ocp, fx, fy = O.post_surface_normal_records(ocl, onr)
        m = ~np.isnan(on)
        ok = len(rec) == len(on) and np.array_equal(np.isnan(rec["normal"]), ~m) and \
            np.array_equal(rec["normal"].view(np.uint32)[m], on.view(np.uint32)[m]) and \
            np.array_equal(rec["camera_position"].view(np.uint32), ocp.view(np.uint32))
        if not ok:
            bad += 1
            print(f"MISMATCH surface normals: frame {i} ({kind})")
    print(f"{n} frames (lines, AHC planes + post-processing, CAPE planes, surface normals each) compared in {time.time() - t0:.0f} s: {bad} mismatches")
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
