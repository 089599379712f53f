#!/usr/bin/env python3
"""Per-call latency of the map-side matchers next to the CPU oracle's, on the synthetic room
sequence: host buffers in, host buffers out, one call at a time.  Not the headline metric; numbers quoted in DESIGN.md."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timeit(f, n):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:      # the GPU drops its clocks while the host builds the scene: wake it up
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from dr_slam_amd import lib, synth
    from dr_slam_amd.pipeline import FrontEnd
    from oracle import oracle as O
    import line_scenarios as LS
    O.lib()
    frames = list(synth.sequence(2, 4))
    cam = synth.TUM3
    fe = FrontEnd(cam, max_batch=8)
    gray = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    depth = torch.from_numpy(np.stack([f[1] for f in frames]).view(np.int16)).cuda()
    Twc = np.stack([f[2] for f in frames]).astype(np.float64)
    Tcw = np.linalg.inv(Twc).astype(np.float32)
    Twc = Twc.astype(np.float32)
    fe.process(gray, depth, Tcw, Twc, th=15.0, check_ori=True, stream=torch.cuda.current_stream().cuda_stream)
    o = O.OrbOracle()
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    of = []
    for g, d, _ in frames:
        kps, desc = o(g)
        of.append(O.FrameOracle(kps, desc, O.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor)), K4, cam.bf, cam.w, cam.h, o.scale))
    rng = np.random.RandomState(7)

    def points(slot, every=True):
        world, valid = of[slot].unproject(Twc[slot])
        valid = valid.astype(bool)
        n = len(world)
        p = np.zeros(n, lib.FRUSTUM_POINT_DTYPE)
        p["world"] = np.where(valid[:, None], world, 0)
        v = Twc[slot][:3, 3][None, :] - p["world"]
        d = np.linalg.norm(v, axis=1) + 1e-9
        p["normal"] = (-(v / d[:, None])).astype(np.float32)
        lvl = of[slot].kps["octave"]
        p["max_distance"] = (d * o.scale[lvl] * 1.1).astype(np.float32)
        p["min_distance"] = (p["max_distance"] / o.scale[-1] * 0.8).astype(np.float32)
        return p, of[slot].desc, (~valid).astype(np.uint8)

    p0, d0, k0 = points(0)
    p1, d1, k1 = points(1)
    inv_sigma2 = fe.ctx.scale_tables()[3]
    T12 = Tcw[0].astype(np.float64) @ Twc[1].astype(np.float64)
    R12, t12 = T12[:3, :3].astype(np.float32), T12[:3, 3].astype(np.float32)
    matched = np.zeros(of[1].N, np.uint8)
    rows = []
    rows.append((f"Fuse(KF, {len(p0)} MapPoints, 3.0) search", timeit(lambda: fe.ctx.fuse_search(1, Tcw[1], p0, d0, k0, 3.0), 50),
                 timeit(lambda: O.fuse_search(of[1], Tcw[1], 1.2, inv_sigma2, p0, d0, k0, 3.0), 5)))
    rows.append((f"SearchBySim3 ({len(p0)} x {len(p1)}, th 7.5)", timeit(lambda: fe.ctx.search_by_sim3(0, 1, Tcw[0], Tcw[1], 1.0, R12, t12, p0, d0, k0, p1, d1, k1, 7.5), 50),
                 timeit(lambda: O.search_by_sim3(of[0], of[1], Tcw[0], Tcw[1], 1.0, R12, t12, 1.2, 8, p0, d0, k0, p1, d1, k1, 7.5), 5)))
    rows.append((f"SearchByProjection(KF, Scw, {len(p0)} points, 10)", timeit(lambda: fe.ctx.search_by_projection_kf(1, Tcw[1], p0, d0, k0, matched, 10.0), 50),
                 timeit(lambda: O.search_by_projection_kf(of[1], Tcw[1], 1.2, 8, p0, d0, k0, matched, 10.0), 5)))
    ang = of[0].kps["angle"]
    rows.append((f"SearchByProjection(Frame, KF, {len(p0)} points, 10, 100) reloc", timeit(lambda: fe.ctx.search_by_projection_reloc(1, Tcw[1], p0, d0, ang, k0, matched, 10.0, 100, True), 50),
                 timeit(lambda: O.search_by_projection_reloc(of[1], Tcw[1], 1.2, 8, p0, d0, ang, k0, matched, 10.0, 100, True), 5)))
    big = np.tile(p0, 6)[:5000]
    camL = lib.Camera(**LS.CAM)
    rows.append(("isInFrustum, 5000 map points", timeit(lambda: fe.ctx.is_in_frustum(Tcw[1], camL, big, 0.5), 50),
                 timeit(lambda: O.is_in_frustum(LS.cam9(), LS.CAM["bf"], Tcw[1], 1.2, 8, big, 0.5), 5)))
    sc = LS.make(11, lib.KEYLINE_DTYPE, lib.MAPLINE_DTYPE, lib.TRACKED_LINE_DTYPE, n_cur=300, n_last=2000)
    lines = np.zeros(2000, lib.FRUSTUM_LINE_DTYPE)
    lines["world"] = sc["last"]["world"]
    mid = 0.5 * (lines["world"][:, :3] + lines["world"][:, 3:])
    Tw = np.linalg.inv(sc["Tcw_cur"].astype(np.float64))
    om = mid - Tw[:3, 3][None, :]
    dist = np.linalg.norm(om, axis=1)
    lines["normal"] = om / dist[:, None]
    lines["max_distance"] = (dist * 1.2).astype(np.float32)
    lines["min_distance"] = (dist * 0.5).astype(np.float32)
    skip = np.zeros(2000, np.uint8)
    rows.append(("LSDmatcher::Fuse(KF, 2000 MapLines) search, 300 key lines", timeit(lambda: fe.ctx.lsd_fuse_search(sc["Tcw_cur"], camL, lines, sc["last"]["desc"], skip, sc["cur"], sc["cur_desc"], 3.0), 50),
                 timeit(lambda: O.lsd_fuse_search(LS.cam9(), sc["Tcw_cur"], 1.2, LS.SCALE, lines, sc["last"]["desc"], skip, sc["cur"], sc["cur_desc"], 3.0), 5)))
    for name, gpu, cpu in rows:
        print(f"{name:68s} product {gpu:7.3f} ms   CPU oracle {cpu:7.3f} ms")
    fe.ctx.close()


if __name__ == "__main__":
    main()
