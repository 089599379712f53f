#!/usr/bin/env python3
"""Latency of the single-frame (host-buffer) paths next to the CPU oracle's: AHC planes, CAPE planes, LSD+LBD lines.
Not the headline metric (bench.py); used for the numbers in DESIGN.md.  Run on a GPU box from the repo root."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(f, n):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from dr_slam_amd import lib, synth
    from oracle import oracle as O
    cam = synth.TUM3
    g, d16, _ = next(synth.sequence(2, 1, kind="room_boxes"))
    ctx = lib.Context(max_batch=1)
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    depth_m = O.depth_to_float(d16, np.float32(inv))
    rows = []
    rows.append(("AHC planes (PlaneDetection)", timeit(lambda: ctx.planes_ahc(d16, K4, inv), 20), timeit(lambda: O.ahc_planes(d16, K4, inv), 5)))
    rows.append(("CAPE planes", timeit(lambda: ctx.planes_cape(depth_m, K4), 20), timeit(lambda: O.cape_planes(depth_m, K4), 5)))
    rows.append(("LSD + LBD lines", timeit(lambda: ctx.lsd_extract(g), 20), timeit(lambda: O.extract_lines(g), 3)))
    rows.append(("ORB extract, host buffers, 1 frame", timeit(lambda: ctx.orb_extract(g), 50), None))
    for name, gpu, cpu in rows:
        print(f"{name:40s} product {gpu:8.2f} ms/frame" + (f"   CPU oracle {cpu:8.2f} ms/frame" if cpu else ""))
    ctx.close()


if __name__ == "__main__":
    main()
