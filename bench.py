#!/usr/bin/env python3
"""bench.py — RGB-D frames/sec (extract+match) at 640x480 on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of `--batch` synthetic frames of one sequence,
already resident in HBM: ORB extraction (pyramid, FAST, quadtree, blur, rBRIEF) -> stereo/grid glue ->
SearchByProjection(frame k, frame k-1) for every consecutive pair of the batch (BASELINE config 2).
N>1: one process per GPU (torch.distributed, backend nccl == RCCL), every rank runs its own sequence —
no data-path collective; timing is barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, HIP-event
timed) and `cpu_baseline` (the CPU oracle timed on this host, single thread, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak

# SURVEY.md §8(d) algorithmic bytes per 640x480 frame, by stage (N = 1000 keypoints)
ALGO_BYTES = {
    "pyramid": 307200 + 1158012 + 926546,     # read gray + write bordered pyramid + read for resize
    "fast": 950532,                           # FAST read of every level interior
    "blur": 1901064,                          # blur read + write
    "desc": 1922000 + 60000,                  # orientation + descriptor patches + outputs
    "match": 676000,                          # window match (a-10)
}


WORKLOAD_TEXT = {
    2: "640x480 synthetic RGB-D (TUM3 intrinsics), ORB extract 1000/1.2/8/20/7 + SearchByProjection(frame k, "
       "frame k-1, th=15) on consecutive frames; BASELINE config 2",
    4: "640x480 synthetic RGB-D, one 256-frame sequence per rank (seed 10+rank, intrinsics cycling TUM1/TUM2/TUM3: "
       "UndistortKeyPoints live on two thirds of the ranks), ORB extract 1000/1.2/8/20/7 + SearchByProjection(frame "
       "k, frame k-1, th=15); BASELINE config 4",
    5: "1280x960 synthetic RealSense-style RGB-D stream (D435 intrinsics x 2, depth factor 1000), 8 pyramid levels, ORB extract "
       "1000/1.2/8/20/7 + SearchByProjection(frame k, frame k-1, th=15), batched on the device; LSD+LBD lines and CAPE planes of "
       "the same frames timed per frame beside it; BASELINE config 5 (a parity / roofline case, not the headline)",
}


def make_batch(base, batch: int, phase: int = 0):
    """Ping-pong over the rendered sequence so every adjacent pair of the batch is a real
    frame-to-frame motion: 0,1,..,n-1,n-2,..,1,0,1,..; phase = how far into that walk the batch starts."""
    from dr_slam_amd.sharding import pingpong_order
    order = pingpong_order(batch + phase, len(base))[phase:]
    gray = np.stack([base[i][0] for i in order])
    depth = np.stack([base[i][1] for i in order])
    Twc = np.stack([base[i][2] for i in order]).astype(np.float64)
    Tcw = np.linalg.inv(Twc)
    return gray, depth, Tcw.astype(np.float32), Twc.astype(np.float32)


def _native_oracle():
    """BASELINE.md section 2 asks for -O3 -march=native -ffp-contract=off: the checked-in oracle build is x86-64-v3 (it has to
    run in the build container too), so the timing leg compiles its own copy for THIS host into a scratch directory and
    points oracle.py at it (never into the tree).  Falls back to the shipped build if the compiler is missing."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle")
    out = os.path.join(tempfile.gettempdir(), "drfe_oracle_native_%d" % os.getuid())
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libdrfe_oracle_native.so")
    # the oracle's own translation units only: ref_shim.cpp wraps sources of /root/reference (oracle/_ref), which does not
    # exist on the GPU box - compiling it there is what made round 3's native rebuild fail, silently
    files = sorted(f for f in os.listdir(src) if f.endswith(".cpp") and f != "ref_shim.cpp")
    try:
        objs = []
        procs = []
        for f in files:
            o = os.path.join(out, f[:-4] + ".o")
            objs.append(o)
            procs.append((f, subprocess.Popen(["g++", "-std=c++17", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-c",
                                               os.path.join(src, f), "-o", o], stderr=subprocess.PIPE, text=True)))
        failed = []
        for f, p in procs:
            err = p.communicate()[1]
            if p.returncode != 0:
                first = next((ln.strip() for ln in (err or "").splitlines() if ln.strip()), "no compiler output")
                failed.append("%s: %s" % (f, first[:160]))
        if failed:
            return None, "-O3 -march=x86-64-v3 (native rebuild failed: %s)" % failed[0]
        link = subprocess.run(["g++", "-shared", "-o", so] + objs, stderr=subprocess.PIPE, text=True)
        if link.returncode != 0:
            first = next((ln.strip() for ln in (link.stderr or "").splitlines() if ln.strip()), "no linker output")
            return None, "-O3 -march=x86-64-v3 (native link failed: %s)" % first[:160]
        return so, "-O3 -march=native"
    except Exception as e:
        return None, "-O3 -march=x86-64-v3 (no compiler on this host: %s)" % str(e)[:120]


def _host_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(base, cam, frames: int = 220, discard: int = 20, only_orb: bool = False):
    """The CPU oracle (kind 'port': the reference cannot be built) timed on this host in the three threading shapes of
    BASELINE.md section 2; steady_clock per frame, first `discard` frames dropped, median and p95:
      (i)   1 thread: ORB extract + SearchByProjection against the previous frame  (= the metric's path; `value`)
      (ii)  the reference's own shape: ORB || LSD+LBD || AHC planes on 3 threads per frame (src/Frame.cc:124-134), then
            the matcher
      (iii) as many independent sequences x 3 threads as the host has cores."""
    from concurrent.futures import ThreadPoolExecutor
    from dr_slam_amd import sharding
    so, flags = _native_oracle()
    if so:
        # take whichever build extracts faster on this host (gcc's -march=native is not always the quicker one)
        import subprocess
        probe = ("import sys,time,numpy as np; sys.path.insert(0, %r); from oracle import oracle as orc; o = orc.OrbOracle(); "
                 "g = (np.arange(640*480, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8).reshape(480, 640); o(g); "
                 "t = time.perf_counter(); [o(g) for _ in range(8)]; print(time.perf_counter() - t)" % ROOT)
        t_native = float(subprocess.run([sys.executable, "-c", probe], env=dict(os.environ, DRFE_ORACLE_LIB=so),
                                        capture_output=True, text=True).stdout or 1e9)
        t_ship = float(subprocess.run([sys.executable, "-c", probe], env={k: v for k, v in os.environ.items() if k != "DRFE_ORACLE_LIB"},
                                      capture_output=True, text=True).stdout or 1e9)
        if t_native < t_ship:
            os.environ["DRFE_ORACLE_LIB"] = so
            flags += " (%.0f %% faster than the shipped x86-64-v3 build here)" % (100 * (t_ship / t_native - 1))
        else:
            flags = "-O3 -march=x86-64-v3 (the -march=native build is %.0f %% slower on this host)" % (100 * (t_native / t_ship - 1))
    from oracle import oracle as orc
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = np.float32(1.0) / np.float32(cam.depth_factor)
    ncpu = sharding.host_cpus()

    def run_sequence(n, full, pool):
        o = orc.OrbOracle()
        prev, ms = None, []
        for i in range(n):
            g, d, Twc = base[i % len(base)]
            t0 = time.perf_counter()
            if full:
                fl = pool.submit(orc.extract_lines, g)
                fp = pool.submit(orc.ahc_planes, d, K4, float(inv))
            kps, desc = o(g)
            fo = orc.FrameOracle(kps, desc, orc.depth_to_float(d, inv), K4, cam.bf, cam.w, cam.h, o.scale, dist=cam.dist)
            if full:
                fl.result(); fp.result()
            Tcw = np.linalg.inv(Twc).astype(np.float32)
            if prev is not None:
                pf, pTwc, pTcw = prev
                world, valid = pf.unproject(pTwc.astype(np.float32))
                mp = np.zeros(pf.N, orc.MAPPOINT_DTYPE)
                mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, pf.desc
                orc.search_by_projection_last(fo, pf, Tcw, pTcw, mp, 15.0, False, True)
            prev = (fo, Twc, Tcw)
            ms.append((time.perf_counter() - t0) * 1e3)
        return np.array(ms[discard:])

    def stats(ms, nseq=1):
        med = float(np.median(ms))
        return {"frames_per_s": nseq * 1e3 / med, "median_ms": med, "p95_ms": float(np.percentile(ms, 95)), "frames": int(len(ms))}

    shapes = {}
    shapes["i_orb_match_1thread"] = dict(stats(run_sequence(frames, False, None)), threads=1)
    n2 = max(discard + 40, frames // 2)                 # the full front-end costs ~3x per frame: bounded sample
    if not only_orb:                                    # AHC is a 640x480 extractor (the reference hard-codes its block grid)
        with ThreadPoolExecutor(2) as pool:
            shapes["ii_orb_lsd_ahc_3threads"] = dict(stats(run_sequence(n2, True, pool)), threads=3)
    nseq = max(1, ncpu // 3)
    if nseq > 1 and not only_orb:
        pools = [ThreadPoolExecutor(2) for _ in range(nseq)]
        with ThreadPoolExecutor(nseq) as outer:
            res = list(outer.map(lambda p: run_sequence(n2, True, p), pools))
        for p in pools:
            p.shutdown()
        shapes["iii_all_cores"] = dict(stats(np.concatenate(res), nseq), threads=3 * nseq, sequences=nseq)
    one = shapes["i_orb_match_1thread"]
    return {"value": one["frames_per_s"], "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{one['frames']} frames {cam.w}x{cam.h} after {discard} discarded (ORB extract + SearchByProjection vs the previous "
                      f"frame), CPU oracle {flags} -ffp-contract=off, 1 thread; 1 / median frame time",
            "host": {"model": _host_model(), "cpus_available": ncpu}, "shapes": shapes}


def config5_aux(ctx, base, cam, n: int = 6):
    """BASELINE config 5 names ORB + LSD + plane: single-frame latency of the line path (device passes + host LSD / LBD), of the
    CAPE plane path and of the AHC plane path (the live extractor, src/Frame.cc:126) at this frame size, on the context the ORB
    batch ran on; and the AHC path + Frame::ComputePlanes' per-plane loop through the batch entry, whose extractor and voxel
    grids run on the device (round 5: 128 x 96 init blocks - the reference hard-codes 64 x 48, include/PlaneExtractor.h:35-36;
    the oracle defines the generalised semantics)."""
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    out = {}
    ctx.lsd_extract(base[0][0])
    t0 = time.perf_counter()
    nl = [len(ctx.lsd_extract(base[i % len(base)][0])["lines"]) for i in range(n)]
    out["lsd_lbd_ms"] = (time.perf_counter() - t0) * 1e3 / n
    dm = [(base[i % len(base)][1].astype(np.float32) * (np.float32(1.0) / np.float32(cam.depth_factor))) for i in range(n)]
    ctx.planes_cape(dm[0], K4, 20)
    t0 = time.perf_counter()
    npl = [len(ctx.planes_cape(d, K4, 20)["planes"]) for d in dm]
    out["cape_ms"] = (time.perf_counter() - t0) * 1e3 / n
    out["lines_per_frame"], out["planes_per_frame"] = float(np.mean(nl)), float(np.mean(npl))
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    d16 = [base[i % len(base)][1] for i in range(n)]
    ctx.planes_ahc(d16[0], K4, inv)
    t0 = time.perf_counter()
    npa = [len(ctx.planes_ahc(d, K4, inv)["planes"]) for d in d16]
    out["ahc_ms"] = (time.perf_counter() - t0) * 1e3 / n
    out["ahc_planes_per_frame"] = float(np.mean(npa))
    res = {}
    for nb in (32, 128, 256):   # a frame is a ~0.3 s chain on one wavefront at this size (12 288 blocks): the rate is frames per call over that latency, until every CU holds a frame (k_ahc_cluster_big: one per CU by LDS)
        db = np.stack([base[i % len(base)][1] for i in range(nb)])
        ctx.planes_ahc_post_batch(db, K4, inv, 5.0, 0.10)                   # Realsense.yaml:76-79; the first call of a size allocates the frame slots
        s0 = ctx.planes_ahc_stats()
        t0 = time.perf_counter()
        _, nn, _, na, _ = ctx.planes_ahc_post_batch(db, K4, inv, 5.0, 0.10)
        el = time.perf_counter() - t0
        s1 = ctx.planes_ahc_stats()
        res[nb] = {"frames": nb, "ms_per_frame": el * 1e3 / nb, "frames_per_s": nb / el, "ms_per_call": el * 1e3, "planes_per_frame": float(nn.mean()),
                   "accepted_per_frame": float(na.mean()), "frames_redone_on_host": s1["to_host"] - s0["to_host"],
                   "voxel_grids_redone_on_host": s1["voxel_grids_to_host"] - s0["voxel_grids_to_host"]}
    # BASELINE config 5 asks for an HBM roofline report of ORB + LSD + plane: the long kernels of the line and plane batch entries at this frame
    # size, each call alone on the device with the kernel clock on; algorithmic bytes = section 8(d)'s per-frame formulas on this geometry
    px = cam.w * cam.h
    algo5 = {"lsd_lbd": px + 2 * (px * 0.64) * 8 + 4 * px, "ahc_planes": 2 * px + px + 0.8 * px}      # gray + (modgrad, angle f64 at 0.8 scale) + Sobel i16 x2; depth u16 + label u8 + member lists
    nb = 128
    gb = np.stack([base[i % len(base)][0] for i in range(nb)])
    db = np.stack([base[i % len(base)][1] for i in range(nb)])
    ctx.lsd_extract_batch(gb, n_threads=2)                                  # arenas
    ctx.long_kernel_clock(True)
    t0 = time.perf_counter()
    lb = ctx.lsd_extract_batch(gb, n_threads=2)
    t_l = time.perf_counter() - t0
    ctx.planes_ahc_post_batch(db, K4, inv, 5.0, 0.10)
    ms = ctx.long_kernel_ms()
    ctx.long_kernel_clock(False)
    out["lsd_batch"] = {"frames": nb, "ms_per_call": t_l * 1e3, "frames_per_s": nb / t_l, "lines_per_frame": float(np.mean([len(a["lines"]) for a in lb])),
                        "frames_returned_to_host": ctx.lsd_stats()}
    out["roofline_long_kernels"] = long_kernel_roofline(ms, nb, algo5)
    out["kernel_ms_one_call_alone"] = {k: round(v, 3) for k, v in ms.items() if v > 0}
    out["algorithmic_bytes_per_frame_lines_planes"] = {k: int(v) for k, v in algo5.items()}
    out["ahc_post_batch"] = dict(res[128], at_32_frames_per_call=res[32], at_256_frames_per_call=res[256],
                                 note="host frames uploaded inside the call; extractor (k_ahc_cluster_big / k_ahc_refine_big, one wavefront per frame, one frame per CU by LDS) + voxel grids + "
                                      "gates and RANSAC refit (k_plane_refit) on the device; the pool uploads, launches and copies the post records")
    return out


def scene_stage_times(B, device):
    """The headline path on the OTHER textures BASELINE's configs name (config 1's namesake is a low-texture scene, config 3 a living
    room): per-stage HIP-event times of one 512-frame batch and the one-context step rate, per scene kind.  The FAST kernel picks
    per cell between its plain path and the screened ones (compass screen at iniThFAST on textured cells, at minThFAST on low-texture ones, + the strength tree on the compacted survivors)."""
    import torch
    from dr_slam_amd import sharding, synth
    from dr_slam_amd.pipeline import FrontEnd
    res = {}
    for kind, cam in (("living_room", synth.ICL), ("planar_lowtexture", synth.TUM3)):
        base = sharding.render_sequence(3, 16, cam, kind, workers=min(8, sharding.host_cpus()))
        order = sharding.pingpong_order(B, len(base))
        gray = torch.from_numpy(np.stack([base[i][0] for i in order])).cuda()
        depth = torch.from_numpy(np.stack([base[i][1] for i in order]).view(np.int16)).cuda()
        Twc = np.stack([base[i][2] for i in order]).astype(np.float64)
        Tcw = np.linalg.inv(Twc).astype(np.float32)
        Twc = Twc.astype(np.float32)
        fe = FrontEnd(cam, max_batch=B, device=device)
        for _ in range(5):
            fe.process(gray, depth, Tcw, Twc, stream=0)
        torch.cuda.synchronize()
        fe.ctx.profile_enable(True)
        acc = {}
        for _ in range(10):
            fe.process(gray, depth, Tcw, Twc, stream=0)
            for k, v in fe.ctx.profile_stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 10
        fe.ctx.profile_enable(False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            fe.process(gray, depth, Tcw, Twc, stream=0)
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        res[kind] = {"stage_ms_per_batch": {k: round(v, 4) for k, v in acc.items()}, "frames_per_s_one_batch_at_a_time": round(B * 50 / el),
                     "keypoints_per_frame_min": int(fe.ctx.orb_counts(B).min())}
        fe.ctx.close()
        del gray, depth
    return res


def host_fed_rate(fe, gray, depth, Tcw, Twc, B, steps, dev):
    """The same step fed from HOST memory: pinned gray + depth copied H2D on a copy stream while the previous batch
    computes (double-buffered inputs), keypoints / descriptors / matches of every batch copied back D2H.  The link, not the
    kernels, bounds this number (472 MB in + 34 MB out per 512 frames)."""
    import torch
    K = fe.ctx.max_kp
    gh, dh = torch.from_numpy(gray).pin_memory(), torch.from_numpy(depth.view(np.int16)).pin_memory()
    gd = [torch.empty_like(gh, device=dev) for _ in range(2)]
    dd = [torch.empty_like(dh, device=dev) for _ in range(2)]
    res = dict(kps=torch.empty((B, K, 28), dtype=torch.uint8).pin_memory(), desc=torch.empty((B, K, 32), dtype=torch.uint8).pin_memory(),
               kc=torch.empty(B, dtype=torch.int32).pin_memory(), m=torch.empty((B, K), dtype=torch.int32).pin_memory(),
               mc=torch.empty(B, dtype=torch.int32).pin_memory())
    s_copy, s_comp = torch.cuda.Stream(), torch.cuda.Stream()
    ev_in = [torch.cuda.Event() for _ in range(2)]        # inputs of buffer k are on the device
    ev_done = [torch.cuda.Event() for _ in range(2)]      # the batch computed from buffer k has been downloaded

    def h2d(k):
        with torch.cuda.stream(s_copy):
            s_copy.wait_event(ev_done[k])
            gd[k].copy_(gh, non_blocking=True)
            dd[k].copy_(dh, non_blocking=True)
            ev_in[k].record(s_copy)

    def run(n):
        for k in range(2):
            ev_done[k].record(s_comp)
        h2d(0)
        for i in range(n):
            k = i & 1
            if i + 1 < n:
                h2d(k ^ 1)
            s_comp.wait_event(ev_in[k])
            fe.process(gd[k], dd[k], Tcw, Twc, th=15.0, check_ori=True, stream=s_comp.cuda_stream)
            fe.ctx.batch_download_async_ptr(B, res["kps"].data_ptr(), res["desc"].data_ptr(), res["kc"].data_ptr(),
                                            res["m"].data_ptr(), res["mc"].data_ptr(), s_comp.cuda_stream)
            ev_done[k].record(s_comp)
        torch.cuda.synchronize()

    run(2)
    t0 = time.perf_counter()
    run(steps)
    el = time.perf_counter() - t0
    assert int(res["kc"].min()) > 500 and int(res["mc"][1:].min()) > 50
    return B * steps / el, el / steps * 1e3


def host_fed_sparse_rate(fes, gray, depth, Tcw, Twc, B, steps, dev, _skip=(), _threads=0):
    """Host-fed with HALF the bytes: only the gray frames cross the link (157 MB per 512 frames); the depth images stay on the
    host, which gathers the one raw value per keypoint the glue reads (drfe_orb_keypoint_pixels_async ->
    drfe_gather_keypoint_depth -> drfe_frame_stereo_grid_batch_kpdepth: 2 MB up, 1 MB down per batch).  Two contexts
    alternate so that the host gather of batch i runs while the device extracts batch i + 1, and three input buffers keep the
    copy of batch i + 2 in flight meanwhile (a copy issued after the gather would add the gather to every step)."""
    import torch
    from dr_slam_amd import sharding
    K = fes[0].ctx.max_kp
    gh = torch.from_numpy(gray).pin_memory()
    gd = [torch.empty_like(gh, device=dev) for _ in range(3)]
    uv = [torch.empty((B, K), dtype=torch.int32).pin_memory() for _ in range(2)]
    kc = [torch.empty(B, dtype=torch.int32).pin_memory() for _ in range(2)]
    kpd = [torch.zeros((B, K), dtype=torch.int16).pin_memory() for _ in range(2)]
    res = [dict(kps=torch.empty((B, K, 28), dtype=torch.uint8).pin_memory(), desc=torch.empty((B, K, 32), dtype=torch.uint8).pin_memory(),
                kc=torch.empty(B, dtype=torch.int32).pin_memory(), m=torch.empty((B, K), dtype=torch.int32).pin_memory(),
                mc=torch.empty(B, dtype=torch.int32).pin_memory()) for _ in range(2)]
    s_copy, s_comp, s_d2h = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ev_in = [torch.cuda.Event() for _ in range(3)]        # input buffer b holds its batch
    ev_ext = [torch.cuda.Event() for _ in range(3)]       # the extraction has read input buffer b: it may be refilled
    ev_uv = [torch.cuda.Event() for _ in range(2)]        # context k: keypoint pixels are on the host
    ev_m = [torch.cuda.Event() for _ in range(2)]         # context k: matched, the results may leave
    ev_done = [torch.cuda.Event() for _ in range(2)]      # ... and have left: context k may extract again
    w, h = gray.shape[2], gray.shape[1]
    threads = _threads or max(1, min(2, sharding.host_cpus() // 4))

    def copy_in(i):      # H2D of batch i into input buffer i % 3
        b = i % 3
        with torch.cuda.stream(s_copy):
            s_copy.wait_event(ev_ext[b])
            if "h2d" not in _skip:
                gd[b].copy_(gh, non_blocking=True)
            ev_in[b].record(s_copy)

    def extract(i):      # extraction + the keypoint pixels of batch i on context i & 1
        b, k = i % 3, i & 1
        s_comp.wait_event(ev_in[b])
        s_comp.wait_event(ev_done[k])
        fes[k].ctx.orb_extract_batch_ptr(gd[b].data_ptr(), w * h, w, w, h, B, s_comp.cuda_stream)
        ev_ext[b].record(s_comp)
        fes[k].ctx.keypoint_pixels_async_ptr(B, uv[k].data_ptr(), kc[k].data_ptr(), s_comp.cuda_stream)
        ev_uv[k].record(s_comp)

    def finish(i):       # host gather, glue from the gathered values, match; the results go home on their own stream
        k = i & 1
        ev_uv[k].synchronize()
        if "gather" not in _skip:
            fes[k].ctx.gather_keypoint_depth(depth, uv[k].numpy().view(np.uint32), kc[k].numpy(), kpd[k].numpy().view(np.uint16), threads)
        fes[k].ctx.stereo_grid_batch_kpdepth_ptr(kpd[k].data_ptr(), True, fes[k].cam, B, s_comp.cuda_stream)
        fes[k].ctx.match_consecutive_batch(Tcw, Twc, fes[k].cam, 15.0, False, True, B, s_comp.cuda_stream)
        ev_m[k].record(s_comp)
        s_d2h.wait_event(ev_m[k])
        r = res[k]
        if "d2h" not in _skip:
            fes[k].ctx.batch_download_async_ptr(B, r["kps"].data_ptr(), r["desc"].data_ptr(), r["kc"].data_ptr(), r["m"].data_ptr(),
                                                r["mc"].data_ptr(), s_d2h.cuda_stream)
        ev_done[k].record(s_d2h)

    def run(n):
        for e in ev_ext + ev_done:
            e.record(s_comp)
        copy_in(0)
        if n > 1:
            copy_in(1)
        extract(0)
        for i in range(n):
            if i + 2 < n:
                copy_in(i + 2)
            if i + 1 < n:
                extract(i + 1)
            finish(i)
        torch.cuda.synchronize()

    run(3)
    t0 = time.perf_counter()
    run(steps)
    el = time.perf_counter() - t0
    if not _skip:
        assert int(res[0]["kc"].min()) > 500 and int(res[0]["mc"][1:].min()) > 50
    return B * steps / el, el / steps * 1e3, threads


# SURVEY.md section 8(d), algorithmic bytes per 640x480 frame of the stages full_frontend adds to the headline path
FF_ALGO_BYTES = {"orb_extract": 7225354, "window_match": 676000, "lsd_lbd": 307200 + 3145728 + 1228800, "ahc_planes": 614400 + 307200 + 245760}


LONG_KERNEL_STAGE = {"k_lsd_order": "lsd_lbd", "k_lsd_grow": "lsd_lbd", "k_rect_improve": "lsd_lbd", "k_ahc_cluster": "ahc_planes", "k_ahc_refine": "ahc_planes",
                     "k_voxel_grid": "ahc_planes", "k_plane_refit": "ahc_planes"}
# SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of the committed counter passes (profiles/r06_long_kernels_summary.txt; a --pmc pass cannot run inside this process)
LONG_KERNEL_LANES = {"k_lsd_grow": 0.729, "k_plane_refit": 0.331, "k_lsd_order": 0.737, "k_rect_improve": 0.083, "k_ahc_cluster": 0.382, "k_ahc_refine": 0.806, "k_voxel_grid": 0.584}


def long_kernel_roofline(ms_by_kernel, frames, algo_bytes=None):
    """per long kernel of the line / plane paths: its stage's algorithmic bytes for the frames of one launch over the
    kernel's duration - measured LIVE in this run by HIP events on the launch stream (drfe_long_kernel_clock: one call alone on the device) -
    against the HBM peak.  These kernels run one wavefront per frame, so the fraction says how far a latency chain is from a streaming pass."""
    ab = algo_bytes or FF_ALGO_BYTES
    out = {}
    for name, ms in ms_by_kernel.items():
        if name in LONG_KERNEL_STAGE and ms > 0:
            ach = ab[LONG_KERNEL_STAGE[name]] * frames / (ms * 1e-3) / 1e9
            out[name] = {"stage": LONG_KERNEL_STAGE[name], "ms_per_launch": ms, "frames_per_launch": frames, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                         "valu_lane_utilisation": LONG_KERNEL_LANES.get(name), "source": "HIP events around the kernel on its launch stream, this run (one call alone on the device)"}
    return out


def full_frontend_roofline(frames_per_s: float, n_frames: int, live_ms=None):
    """HBM roofline of what bounds BASELINE config 3: the aggregate (section 8(d)'s algorithmic bytes of all four stages x the measured
    rate, against 8 TB/s) and, per long kernel of the line and plane paths, long_kernel_roofline of the durations measured in this run."""
    per_frame = sum(FF_ALGO_BYTES.values())
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_frame": per_frame,
           "achieved": per_frame * frames_per_s / 1e9, "frac": per_frame * frames_per_s / 1e9 / HBM_PEAK_GBS,
           "limited_by": "latency chains: every frame's region growing / plane clustering / flood fill is an order-defined sequence on one or four wavefronts; "
                         "the rate is (frames resident) / (chain latency), and residency is bounded by LDS (a frame's `used` bitmap, queues, tables)"}
    out["long_kernels"] = long_kernel_roofline(live_ms or {}, n_frames)
    out["long_kernels_lane_utilisation_source"] = "profiles/r06_long_kernels_summary.txt (rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU)"
    return out


def full_frontend(cam_name, n_frames: int = 512, reps: int = 6, inflight: int = 5, device: int = 0, ranks_on_host: int = 1, sync=None):
    """BASELINE config 3 (ICL-NUIM living-room style, ICL intrinsics): the whole per-frame front-end - ORB + glue +
    SearchByProjection and the surface normals batched on the device; LSD + LBD lines with the detector's sequential core on
    the device (pixel ordering = std::sort's permutation, region growing, rectangle fit / refinement: one wavefront per frame);
    AHC planes with PEAC's extractor on the device (graph, agglomerative clustering, flood fill, re-merge, labels: one
    wavefront per frame) and their PCL-style post-processing on host threads; CAPE planes per frame; NFA arithmetic and key
    lines of the line path on host threads.
    A frame's region growing / plane extraction is a dependency chain of 0.1-0.25 s on ONE wavefront, so the device paths run at
    (frames in flight) / (that latency): 512 frames per step, and `inflight` steps at a time (each on its own contexts) so that
    the host stages of one step run while the other step's wavefronts are on the device.
    Sharded (`bench.py --gpus N --full-frontend`): every rank runs this on its own device with its share of the host's CPUs
    (`ranks_on_host`), `sync` = the ranks' barrier, called right before and right after the timed region; the caller reduces
    (frames, seconds) over the ranks - whole sequences per rank, no data-path collective."""
    import threading
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from dr_slam_amd import lib, sharding, synth
    from dr_slam_amd.pipeline import FrontEnd
    cam = getattr(synth, cam_name)
    ncpu = sharding.host_cpus() // max(1, ranks_on_host)
    if ncpu < 3:                                      # 2.3 cores busy per GPU at the measured rate: with fewer the host, not the device, is measured
        raise RuntimeError("full_frontend: %d host CPUs per rank - the whole front-end of config 3 keeps 2.5 busy per GPU (upload threads, pools, Python)" % ncpu)
    base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
    order = sharding.pingpong_order(n_frames, len(base))
    gray = np.stack([base[i][0] for i in order])
    depth = np.stack([base[i][1] for i in order])
    Twc = np.stack([base[i][2] for i in order]).astype(np.float64)
    Tcw = np.linalg.inv(Twc).astype(np.float32)
    Twc = Twc.astype(np.float32)
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    depth_m = depth.astype(np.float32) * np.float32(inv)
    # the host images the batch entries upload live in PINNED memory, as a capture pipeline's ring buffers would: a copy from
    # pageable memory is staged by a runtime thread first (1.1 GB per step here: 0.2-0.3 ms of CPU per frame that is not the
    # front-end's)
    gray, depth, depth_m = (torch.from_numpy(a).pin_memory().numpy() for a in (gray, depth, depth_m))
    gray_t = torch.from_numpy(gray).to("cuda:%d" % device)
    depth_t = torch.from_numpy(depth.view(np.int16)).to("cuda:%d" % device)
    inflight = max(1, int(os.environ.get("DRFE_FF_INFLIGHT", inflight)))
    # host threads per step in flight.  Round 4: the line path's threads only upload the frames, launch, and copy the finished key
    # lines out (ordering, growth, rect_improve / NFA, key lines and LBD all run on the device): three of them; the plane pool
    # still runs gates + RANSAC refit on the voxel clouds the device left (~0.6 ms per frame), CAPE ~0.2 ms
    nthr = max(2, (ncpu * 26 + 5 * inflight) // (10 * inflight))
    split = {"lines": 2}
    split["planes"] = max(1, min(6, nthr - split["lines"]))
    if os.environ.get("DRFE_FF_SPLIT"):              # experiments: "lines,planes" per step in flight
        split["lines"], split["planes"] = (int(v) for v in os.environ["DRFE_FF_SPLIT"].split(","))
    n_cape = int(os.environ.get("DRFE_FF_CAPE", 2))   # CAPE lanes (host threads of drfe_planes_cape_batch)
    lanes = []
    for _ in range(inflight):
        lanes.append({"fe": FrontEnd(cam, max_batch=n_frames, device=device), "planes": lib.Context(max_batch=1, device=device),
                      "cape": lib.Context(max_batch=1, device=device), "wall": {}, "stream": torch.cuda.Stream(device)})

    def step(L, pool):
        wall = L["wall"]

        def timed(name, fn):
            t = time.perf_counter()
            r = fn()
            wall[name] = (time.perf_counter() - t) * 1e3
            return r

        def planes():       # AHC planes + Frame::ComputePlanes' per-plane loop
            _, n, _, na, _ = L["planes"].planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=split["planes"])
            return len(n), int(na.sum())

        fl = pool.submit(timed, "lines", lambda: L["fe"].ctx.lsd_extract_batch(gray, n_threads=split["lines"]))
        fp = pool.submit(timed, "ahc_planes", planes)
        fc = pool.submit(timed, "cape", lambda: int(L["cape"].planes_cape_batch(depth_m, K4, 20, n_threads=n_cape)[1].sum()))
        t = time.perf_counter()
        st = L["stream"]
        with torch.cuda.stream(st):
            L["fe"].process(gray_t, depth_t, Tcw, Twc, th=15.0, check_ori=True, stream=st.cuda_stream)
            L["fe"].ctx.surface_normals_batch_ptr(depth_t.data_ptr(), cam.w * cam.h, cam.w, cam.w, cam.h, K4, inv, 9.0, n_frames, st.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(st)
        while not ev.query():                        # sleep, not spin: the step's pools need the CPUs
            time.sleep(0.0005)
        wall["orb_match_normals_device"] = (time.perf_counter() - t) * 1e3
        nl, (npl, nacc), ncp = len(fl.result()), fp.result(), fc.result()
        assert nl == n_frames and npl == n_frames and ncp > 0
        return nacc

    nacc = [0] * inflight
    with ThreadPoolExecutor(4 * inflight) as pool:
        for L in lanes:                              # warm-up, one lane at a time: arenas, pinned buffers, graphs
            step(L, pool)
            step(L, pool)

        # every long kernel's duration, live: one call of each path ALONE on the device with the kernel clock on (HIP events on the launch stream)
        L0 = lanes[0]
        L0["fe"].ctx.long_kernel_clock(True); L0["planes"].long_kernel_clock(True)
        L0["fe"].ctx.lsd_extract_batch(gray, n_threads=split["lines"])
        L0["planes"].planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=split["planes"])
        live_ms = dict(L0["fe"].ctx.long_kernel_ms())
        live_ms.update({k: v for k, v in L0["planes"].long_kernel_ms().items() if v > 0})
        L0["fe"].ctx.long_kernel_clock(False); L0["planes"].long_kernel_clock(False)
        step_ms, step_back = [], []

        def handed_back(L):                           # frames the device handed back to the host so far, by stage (cumulative counters of the lane's contexts)
            a = L["fe"].ctx.lsd_stats()
            b = L["planes"].planes_refit_stats()
            c = L["planes"].planes_ahc_stats()
            return np.array([a["grow_to_host"], a["nfa_to_host"], a["keylines_to_host"], b["to_host"], c["to_host"], c["voxel_grids_to_host"]], np.int64)

        def run(k):
            for _ in range(reps):
                h0, t = handed_back(lanes[k]), time.perf_counter()
                nacc[k] = step(lanes[k], pool)
                step_ms.append((time.perf_counter() - t) * 1e3)
                step_back.append(handed_back(lanes[k]) - h0)

        th = [threading.Thread(target=run, args=(k,)) for k in range(inflight)]
        import ctypes
        pool_ns = (ctypes.c_longlong * 3)()
        dbg = getattr(lanes[0]["planes"].L, "drfe_debug_pool_cpu_ns", None)     # measurement hook: CPU time of the pools' threads
        if dbg is not None:
            dbg.restype = None
            dbg(pool_ns)                                                       # clear
        if sync:
            sync()
        cpu0 = time.process_time()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        if sync:
            sync()
        el = time.perf_counter() - t0
        cpu = time.process_time() - cpu0
        if dbg is not None:
            dbg(pool_ns)
    lsd = {"frames": 0, "grow_to_host": 0, "nfa_to_host": 0, "keylines_to_host": 0}
    refit = {"frames": 0, "to_host": 0}
    for L in lanes:
        for k, v in L["fe"].ctx.lsd_stats().items():
            lsd[k] += v
        for k, v in L["planes"].planes_refit_stats().items():
            refit[k] += v
    for L in lanes:
        L["planes"].close(); L["cape"].close(); L["fe"].ctx.close()
    total = inflight * reps * n_frames
    return {"workload": "BASELINE config 3: living_room scene, ICL intrinsics, 640x480: ORB + glue + SearchByProjection + surface "
                        "normals batched on the device; LSD+LBD lines, AHC planes + post-processing, CAPE planes for every frame",
            "value": total / el, "unit": "frames/s", "frames_timed": total, "seconds_timed": el, "frames_per_step": n_frames, "steps_in_flight": inflight, "steps_timed": inflight * reps,
            "ms_per_step": el * 1e3 / (inflight * reps),
            "host_cpu_ms_per_frame": cpu * 1e3 / total, "host_cpu_utilisation": cpu / (el * ncpu),
            "host_cpu_ms_per_frame_by_pool": {"lines": pool_ns[0] / 1e6 / total, "ahc_planes": pool_ns[1] / 1e6 / total, "cape": pool_ns[2] / 1e6 / total,
                                              "other (python, HIP runtime threads)": (cpu * 1e3 - sum(pool_ns) / 1e6) / total},
            "host_threads_per_step_in_flight": {"lines": split["lines"], "ahc_planes": split["planes"], "cape": n_cape}, "host_cpus_available": ncpu,
            "host_cores_busy_at_this_rate": round(cpu / el, 2),
            "host_cores_busy_at_8_gpus_at_this_rate": round(8 * cpu / el, 1),
            "lines_frames_returned_to_the_host": {"of_frames": lsd["frames"], "region_growing": lsd["grow_to_host"], "nfa_decisions_not_certified": lsd["nfa_to_host"],
                                                  "keyline_roundings_not_certified": lsd["keylines_to_host"]},
            "planes_frames_sent_to_the_host_refit": {"of_frames": refit["frames"], "refit_not_certified_or_grid_returned": refit["to_host"]},
            "stage_wall_ms_last_step": {k: round(v, 2) for k, v in lanes[0]["wall"].items()},
            # the tail: wall time of one step (512 frames through every stage, `steps_in_flight` of them side by side) and what the device handed
            # back to the host inside it - a frame whose refit / voxel grid / NFA decision could not be certified is redone on the pool's threads
            # and the whole batch call waits for it
            "step_wall_ms": {"p50": float(np.percentile(step_ms, 50)), "p95": float(np.percentile(step_ms, 95)), "p99": float(np.percentile(step_ms, 99)),
                             "max": float(np.max(step_ms)), "steps": len(step_ms)},
            "frames_handed_back_per_step": {k: {"mean": float(v.mean()), "max": int(v.max())} for k, v in zip(
                ("lines_region_growing", "lines_nfa", "lines_keylines", "planes_refit", "planes_extractor", "planes_voxel_grid"), np.array(step_back).T)},
            "step_wall_ms_with_and_without_hand_back": (lambda b, m: {"steps_with": int((b > 0).sum()), "median_with": float(np.median(m[b > 0])) if (b > 0).any() else None,
                                                                       "median_without": float(np.median(m[b == 0])) if (b == 0).any() else None})(
                np.array(step_back).sum(1), np.array(step_ms)),
            "roofline": full_frontend_roofline(total / el, n_frames, live_ms),
            "kernel_ms_one_call_alone": {k: round(v, 3) for k, v in live_ms.items() if v > 0},
            "planes_accepted_per_step": int(nacc[0]),
            "lines_path": "everything on the device: pixel ordering, region growing, region2rect, refine (k_lsd_order, k_lsd_grow: one wavefront per frame at this call size; k_lsd_grow_mw, four per frame, up to 256 frames per call), rect_improve + NFA "
                          "decisions with certified comparisons (k_rect_improve), key lines + the response cut + line equations (k_lsd_keylines), LBD (k_lbd); host threads upload, launch and copy",
            "planes_path": "init-block fits, graph, agglomerative clustering, flood fill, re-merge, labels, plane clouds (k_ahc_blocks, k_ahc_cluster + k_ahc_refine: one wavefront per frame) and "
                           "pcl::VoxelGrid of every plane (k_voxel_grid) and, round 5, the gates + RANSAC / least-squares refit of Frame::MaxPointDistanceFromPlane (k_plane_refit: one wavefront per plane) "
                           "on the device; the pool's threads upload, launch and copy 24-byte post records",
            "note": "the device paths are latency chains (~0.12 s of region growing, ~0.075 s of plane extraction per frame on one wavefront), so steps run side by side; "
                    "what bounds the rate is LDS x time: every long kernel holds 20-38 KB of a CU's 160 KB for as long as it runs, and the sum over a frame's "
                    "kernels of (LDS held x time held) against the device's 40 MB of LDS is the step time measured (DESIGN.md, what bounds the full front-end).  CAPE and, since round 5, the planes' gates + RANSAC refit run on the device: no per-frame host stage remains "
                    "(host_cpu_ms_per_frame_by_pool is the measurement; host_cores_busy_at_8_gpus_at_this_rate = 8 x the busy cores measured here)"}


def per_frame_latency(cam_name, frames: int = 240, discard: int = 20):
    """BASELINE config 3 in the shape Frame::Frame has (reference src/Frame.cc:124-134): ONE frame at a time, its three extractors
    side by side on three threads and joined - ORB + the Frame glue through drfe_frame_submit / drfe_frame_collect (device), the
    line segment detector through drfe_lsd_extract and the AHC planes + Frame::ComputePlanes' per-plane loop through drfe_planes_ahc
    + drfe_planes_ahc_postprocess (device image passes / block fits, the sequential cores on the calling threads: the low-latency
    entries).  Wall time per frame from the first call to the last join, host frames in, host results out; median and p95.
    The CPU oracle in the same shape is cpu_baseline.shapes.ii_orb_lsd_ahc_3threads."""
    from concurrent.futures import ThreadPoolExecutor
    from dr_slam_amd import lib, sharding, synth
    cam = getattr(synth, cam_name)
    base = sharding.render_sequence(3, 16, cam, "living_room", workers=1)
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    cc = lib.make_camera(cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.depth_factor, cam.w, cam.h)
    orb, lines, planes = lib.Context(max_batch=2), lib.Context(max_batch=1), lib.Context(max_batch=1)
    ms, parts = [], {"orb_glue": [], "lsd_lbd": [], "ahc_planes_post": []}

    def timed(key, fn):
        t = time.perf_counter()
        r = fn()
        parts[key].append((time.perf_counter() - t) * 1e3)
        return r

    def do_planes(d):
        a = planes.planes_ahc(d, K4, inv)
        return planes.planes_ahc_postprocess(d, K4, inv, a, 9.0, 0.10)["n_accepted"]

    try:
        with ThreadPoolExecutor(2) as pool:
            for k in range(frames + discard):
                g, d, _ = base[k % len(base)]
                t0 = time.perf_counter()
                fl = pool.submit(timed, "lsd_lbd", lambda: len(lines.lsd_extract(g)["lines"]))
                fp = pool.submit(timed, "ahc_planes_post", lambda: do_planes(d))
                t1 = time.perf_counter()
                orb.frame_submit(k & 1, g, d, cc)
                kps = orb.frame_collect(k & 1, stereo=True)[0]
                parts["orb_glue"].append((time.perf_counter() - t1) * 1e3)
                nl, na = fl.result(), fp.result()
                ms.append((time.perf_counter() - t0) * 1e3)
                assert len(kps) > 500 and nl > 5 and na >= 1
    finally:
        orb.close(); lines.close(); planes.close()
    a = np.array(ms[discard:])
    return {"shape": "Frame::Frame (src/Frame.cc:124-134): ORB + glue || LSD + LBD || AHC planes + per-plane post-processing, three threads joined, one frame at a time, host in / host out",
            "frames": int(len(a)), "median_ms": float(np.median(a)), "p95_ms": float(np.percentile(a, 95)), "frames_per_s_one_sequence": float(1e3 / np.median(a)),
            "median_ms_by_thread": {k: float(np.median(v[discard:])) for k, v in parts.items()},
            "entries": "drfe_frame_submit / drfe_frame_collect; drfe_lsd_extract; drfe_planes_ahc + drfe_planes_ahc_postprocess",
            "note": "the single-frame entries keep the detectors' sequential cores on the calling threads: one frame's region growing is a 45 ms chain on four "
                    "wavefronts (75 on one) against 13 ms on one host core, its plane clustering + flood fill 48 ms against 3.4 ms - the device wins by "
                    "holding hundreds of frames at once (full_frontend), not one.  Compare with cpu_baseline.shapes.ii_orb_lsd_ahc_3threads."}


def launch(args) -> int:
    """`--gpus N` without a launcher around us: start N fresh rank processes (one per GPU) BEFORE this process makes
    any GPU call - the parent never initialises HIP, it only counts devices - and pass rank 0's JSON line through.
    Non-zero exit if any rank fails.  On a box with fewer than N devices the ranks share device 0 over gloo
    (rehearsal of the N-rank control path; flagged in the JSON line, not a scaling measurement)."""
    import socket
    import subprocess
    import torch
    n = args.gpus
    from dr_slam_amd import sharding
    # every rank needs a host core of its own to drive its GPU (render its sequence, submit, collect); the whole front-end of
    # config 3 needs 2.3 more per GPU for the planes' gates + RANSAC refit (full_frontend.host_cores_busy_at_this_rate)
    if sharding.host_cpus() < n:
        sys.stderr.write("bench.py --gpus %d: only %d host CPUs are available to this job - one per rank is the least the sharded mode needs "
                         "(2.5 per GPU for the whole front-end of config 3); refusing to report a number bound by the host\n" % (n, sharding.host_cpus()))
        return 3
    if getattr(args, "full_frontend", False) and sharding.host_cpus() < 3 * n:
        sys.stderr.write("bench.py --gpus %d --full-frontend: %d host CPUs are available to this job and the whole front-end of config 3 keeps ~2.5 busy "
                         "per GPU (upload threads, pools, Python): 3 per rank is the least; refusing to report a number bound by the host\n" % (n, sharding.host_cpus()))
        return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:
        env["DRFE_BENCH_ONE_DEVICE"] = "1"
        env["DRFE_BENCH_BACKEND"] = "gloo"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's output is read on a thread while ALL children are polled: a rank that dies during start-up leaves the others
    # waiting in the rendezvous / first collective, so the first non-zero exit (or the deadline) ends the whole job
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    deadline = time.time() + 1800
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc = 124
            break
        time.sleep(0.2)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    out0 = b"".join(chunks).decode()
    if rc:
        sys.stderr.write(out0)
        return rc
    sys.stdout.write(out0)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 1000 timed steps (~2.3 s of device time): the steady rate.  Shorter runs are not wrong, but they sit on a clock transient of
    # the device - at batch 512 one context does 203 k frames/s over 20 steps, 192 k over 200 and 203-206 k over 1000; three in
    # flight 220 k / 213 k / 225 k (three fresh processes each, one box)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (0 = 512; 128 at config 5, whose frames have 4x the pixels)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent batches in flight per GPU: consecutive steps alternate between this many contexts, each on the "
                         "stream its context owns, so that the latency-bound kernels of one batch (quadtree, claim resolution, "
                         "the small glue kernels: 40 %% VALU-busy or less) run beside the VALU-bound ones of the others.  Measured "
                         "(fresh processes, 120 steps): 1 x 512: 203 k frames/s, 2 x 512: 222 k, 3 x 512: 227-231 k, 4 x 512: 218-223 k "
                         "(more contexts than hardware queues).  The contexts are a drfe_pipeline object of the C-ABI")
    ap.add_argument("--config", type=int, default=0, choices=(0, 2, 4, 5),
                    help="BASELINE.json config: 2 = TUM3 single sequence, 4 = TUM1/2/3 mix, one 256-frame sequence "
                         "per rank; 5 = 1280x960 RealSense-style stream (roofline report at 4x the pixels); 0 = config 2 at one "
                         "rank, config 4 at N > 1")
    ap.add_argument("--distinct", type=int, default=0,
                    help="frames rendered per rank (0 = the config's sequence length: 64 / 256)")
    ap.add_argument("--render-workers", type=int, default=0,
                    help="processes that render the synthetic sequence (0 = the CPUs this rank may use; forced to 1 under a "
                         "profiler, whose preloaded runtime must not be forked)")
    ap.add_argument("--preload", type=int, default=400,
                    help="single-context steps measured BEFORE the timed loop (reported as one_batch_at_a_time; they also carry the "
                         "device through its clock transient under load); 0 under a profiler")
    ap.add_argument("--settle", type=int, default=300,
                    help="untimed steps of the TIMED loop's own shape (all contexts in flight) before the warm-up: the device's clocks and the contexts the "
                         "single-context preload never touched settle under the load the timed region will apply (measured: 20 timed steps straight after "
                         "the preload 2.06-2.10 ms per step, after 300 such steps 1.92; a 1000-step run 1.88); 0 under a profiler")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-fed rate and the config-3 full front-end block")
    ap.add_argument("--bow", action="store_true", help="also run the vocabulary tree descent in every step")
    ap.add_argument("--full-frontend", action="store_true",
                    help="at --gpus N > 1: after the headline, every rank also runs BASELINE config 3's whole front-end (ORB + LSD + AHC + CAPE + match) on "
                         "its own device and sequence; rank 0 reports total frames over the slowest rank's time.  Needs 3 host CPUs per rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is used\n")
    config = args.config or (2 if world == 1 else 4)

    # Render this rank's sequence first: the pool forks, so it must run before anything touches the GPU.
    from dr_slam_amd import sharding
    seed, cam, kind, seq_len = sharding.rank_workload(config, rank)
    n_distinct = args.distinct or seq_len
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    profiled = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) \
        or any(k.startswith("ROCPROF") for k in os.environ)
    workers = 1 if profiled else (args.render_workers or max(1, sharding.host_cpus() // local_world))
    t_r0 = time.perf_counter()
    base = sharding.render_sequence(seed, n_distinct, cam, kind, workers=workers)
    t_render = time.perf_counter() - t_r0

    import torch
    import torch.distributed as dist

    # Rehearsal hooks for a box with fewer devices than ranks (set by launch(); the driver never sets them):
    # DRFE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and DRFE_BENCH_BACKEND=gloo replaces RCCL, which refuses
    # two ranks on one device.
    rehearsal = bool(os.environ.get("DRFE_BENCH_ONE_DEVICE"))
    if rehearsal:
        local_rank = 0
    backend = os.environ.get("DRFE_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from dr_slam_amd.pipeline import FrontEnd

    B = args.batch or (128 if config == 5 else 512)
    gray, depth, Tcw, Twc = make_batch(base, B)
    nfl = max(1, args.inflight)
    # Batches in flight are a library object (include/drfe.h drfe_pipeline_*): nfl contexts used round robin, each on the stream
    # it owns - those sit on different hardware queues.  (Streams handed out by torch's pool did not, in a fresh process: the
    # batches then queue behind each other and nothing overlaps - tools/streams_in_flight_probe.py, 207 k against 227 k frames/s.)
    from dr_slam_amd import lib as drfe_lib
    pipe = drfe_lib.Pipeline(nfl, 1000, 1.2, 8, 20, 7, cam.w, cam.h, B, local_rank)
    fes = [FrontEnd(cam, max_batch=B, device=local_rank, ctx=c) for c in pipe.contexts]
    fe = fes[0]
    # One resident input batch PER CONTEXT, each starting elsewhere in the sequence's ping-pong walk: consecutive steps read
    # different frames (nfl x 472 MB of gray + depth, more than the 256 MB Infinity Cache holds; round 3 fed every step the same
    # 512 frames, whose gray half partly stayed in that cache)
    phases = [k * 11 for k in range(nfl)]
    inputs = [(gray, depth, Tcw, Twc)] + [make_batch(base, B, ph) for ph in phases[1:]]
    gray_ts = [torch.from_numpy(g).to(dev) for g, _, _, _ in inputs]
    depth_ts = [torch.from_numpy(d.view(np.int16)).to(dev) for _, d, _, _ in inputs]
    gray_t, depth_t = gray_ts[0], depth_ts[0]
    next_ctx = [0]
    torch.cuda.synchronize()                           # the inputs are in HBM before any context's stream reads them

    exchange = None
    if world > 1 or args.bow:
        t_x = time.perf_counter()
        # the one initial exchange of the sharded mode: rank 0 owns the ORB vocabulary
        # (k=10, L=6, ~50 MB flattened; synthetic because the reference's ORBvoc blob is missing) and
        # broadcasts it over RCCL/xGMI; every rank uploads its copy into its own context(s).
        from dr_slam_amd import vocabulary as V
        if rank == 0:
            blob = V.make_synthetic(10, 6, seed=1).pack()
            size = np.array([blob.size], np.int64)
        else:
            size = np.zeros(1, np.int64)
        size = sharding.broadcast_tables(size, dev, dist if world > 1 else None)
        if rank != 0:
            blob = np.zeros(int(size[0]), np.uint8)
        blob = sharding.broadcast_tables(blob, dev, dist if world > 1 else None)
        torch.cuda.synchronize()
        exchange = {"collective": "broadcast of the ORB vocabulary from rank 0 (%s)" % ("RCCL" if backend == "nccl" else backend),
                    "world_size_seen_by_the_collective": dist.get_world_size() if world > 1 else 1, "bytes": int(blob.size),
                    "ms": round((time.perf_counter() - t_x) * 1e3, 2), "data_path_collectives": 0}
        voc = V.Vocabulary.unpack(blob)
        for f in (fes if args.bow else fes[:1]):
            voc.upload(f.ctx)

    def step(k=None):
        """One batch through the hot path: on the pipeline's next context (default) or on context k's own stream."""
        if k is None:
            j = next_ctx[0]                              # the context drfe_pipeline_submit takes next: round robin
            k = pipe.submit(gray_ts[j].data_ptr(), depth_ts[j].data_ptr(), cam.w * cam.h, cam.w, cam.w, cam.h, inputs[j][2], inputs[j][3], fe.cam, 15.0, False, True, B)
            assert k == j, (k, j)
            next_ctx[0] = (j + 1) % nfl
        else:
            fes[k].process(gray_ts[k], depth_ts[k], inputs[k][2], inputs[k][3], th=15.0, check_ori=True, stream=0)
        if args.bow:   # Frame::ComputeBoW tree descent for every frame of the batch (not part of the metric)
            fes[k].ctx.bow_transform_batch(4, B, 0)

    # Before the timed loop, on every rank: the same batch ONE at a time (a single context, nothing overlaps; 400 steps) and the
    # per-stage times by HIP events (10 steps).  Both are reported - and they put about a second of load on the device first: its
    # clocks go through a transient in the first half second under load (one context: 203 k frames/s over 20 steps, 192 k over
    # 200, 203-206 k over 1000), which a 20-step timed loop straight after start-up may or may not hit (200-236 k measured).
    one_at_a_time = None
    if profiled:                                    # a kernel trace of the timed steps, not of the pre-steps
        args.preload, args.settle = 0, 0
    n1 = max(0, args.preload)
    if n1:
        for _ in range(5):
            step(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n1):
            step(0)
        torch.cuda.synchronize()
        one_at_a_time = {"value": B * n1 / (time.perf_counter() - t1), "unit": "frames/s", "steps": n1,
                         "note": "one context, one batch at a time (measured before the timed loop): what the sum of the per-stage "
                                 "kernel times corresponds to"}
    # per-stage kernel time: HIP events around every stage on the launch stream
    fe.ctx.profile_enable(True)
    acc = {}
    reps = max(3, min(args.steps, 10))
    for _ in range(reps):
        step(0)                                     # one context, one stream: clean per-kernel times
        ms = fe.ctx.profile_stage_ms()
        for k, v in ms.items():
            acc[k] = acc.get(k, 0.0) + v
    fe.ctx.profile_enable(False)

    for _ in range(max(0, args.settle)):            # untimed, the timed loop's own load pattern
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    el, total_frames = sharding.reduce_elapsed_and_frames(el, B * args.steps, dev, dist if world > 1 else None)

    # sanity: the batch really produced keypoints and matches
    for f in fes:
        counts = f.ctx.orb_counts(B)
        _, nm = f.matches(B - 1)
        if not os.environ.get("DRFE_BENCH_NO_SANITY"):      # kernel experiments with deliberately wrong results
            assert counts.min() > 500 and nm > 100, (counts.min(), nm)

    # Parity at the bench's own size (outside the timed region): the contexts still hold the last batches of the timed loop -
    # batch-512 results produced with `nfl` batches in flight.  16 random slots of every context are compared with the CPU
    # oracle (the checker, never the thing measured): keypoint records and descriptors of slot s and s - 1, match array of s.
    # A mismatch fails the run.
    parity_checked = 0
    if not os.environ.get("DRFE_BENCH_NO_SANITY"):
        from oracle.spot_check import SlotChecker
        from dr_slam_amd.sharding import pingpong_order
        checker = SlotChecker(cam)
        rng = np.random.default_rng(1234 + rank)
        for ci, f in enumerate(fes):
            keys = pingpong_order(B + phases[ci], len(base))[phases[ci]:]
            slots = np.sort(rng.choice(np.arange(1, B), size=min(16, B - 1), replace=False))
            g_c, d_c, Tcw_c, Twc_c = inputs[ci]
            parity_checked += checker.check(f, g_c, d_c, Tcw_c, Twc_c, slots, keys=keys, th=15.0, check_ori=True,
                                            what="rank %d context %d " % (rank, ci))

    out = None
    if rank == 0:
        stage_ms = {k: v / reps for k, v in acc.items()}
        # FAST runs as two launches (k_fast_cells_cols<8> over the cells of <= 8 rows per lane, <12> over the rest): each has its
        # own event pair and its share of the FAST read (every pixel of the detection region belongs to exactly one cell)
        cells, px = fe.ctx.fast_partition(cam.w, cam.h)
        algo_bytes = dict(ALGO_BYTES)
        if (cam.w, cam.h) != (640, 480):
            # SURVEY.md section 8(d)'s per-frame figures are functions of the level sizes: the same formulas on this geometry
            lv = [fe.ctx.pyramid_level(0, l).shape for l in range(fe.ctx.nlevels)]          # bordered (h + 38, w + 38)
            interior = [(h - 38) * (w - 38) for h, w in lv]
            algo_bytes["pyramid"] = cam.w * cam.h + sum(h * w for h, w in lv) + sum(interior[:-1])
            algo_bytes["fast"] = sum(interior)
            algo_bytes["blur"] = 2 * sum(interior)
        tot = float(px.sum())
        fast_all = algo_bytes["fast"]
        algo_bytes["fast"] = fast_all * float(px[0]) / tot
        algo_bytes["fast_b"] = fast_all * float(px[1]) / tot
        cand = {k: stage_ms[k] for k in algo_bytes if stage_ms.get(k, 0.0) > 0}
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        roof_stage = dom if dom in cand else max(cand, key=lambda k: cand[k])
        algo = algo_bytes[roof_stage] * B
        achieved = algo / (stage_ms[roof_stage] * 1e-3) / 1e9
        # HBM traffic and VALU utilisation of that kernel from the committed PMC passes (rocprofv3 --pmc, separate runs;
        # cannot be collected from inside this process). Only valid for the same batch size.
        traffic, traffic_src, valu, valu_cyc = None, None, None, 4.0
        kname = {"pyramid": "k_pyr_resize_lds", "fast": "k_fast_cells_cols<8>" if cells[1] else "k_fast_cells",
                 "fast_b": "k_fast_cells_cols<12>", "blur": "k_blur", "desc": "k_orient_desc",
                 "match": "k_window_candidates"}[roof_stage]
        try:
            # the newest committed counter passes of this configuration and batch size
            cands = ["r05_pmc_traffic_c5_b%d.json", "r04_pmc_traffic_c5_b%d.json", "r02_pmc_traffic_c5_b%d.json"] if config == 5 else \
                    ["r06_pmc_traffic_b%d.json", "r05_pmc_traffic_b%d.json", "r04_pmc_traffic_b%d.json", "r03_pmc_traffic_b%d.json", "r02_pmc_traffic_b%d.json"]
            pmc_name = next((c % B for c in cands if os.path.exists(os.path.join(ROOT, "profiles", c % B))), cands[-1] % B)
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            if pmc.get("batch") == B and kname in pmc["kernels"] and roof_stage != "pyramid":
                k = pmc["kernels"][kname]
                traffic = float(k["HBM_BYTES_per_launch"])
                traffic_src = "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE: the gfx950 half-count correction, calibrated, see its _about)" % pmc_name
                valu = k.get("valu_issue_utilisation")
                valu_cyc = k.get("cycles_per_valu_instruction", 4.0)
        except Exception:
            pass
        fps = total_frames / el
        out = {
            "metric": "RGB-D frames/sec (extract+match) at %dx%d" % (cam.w, cam.h),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": WORKLOAD_TEXT[config], "baseline_config": config, "batch_per_gpu": B,
                       "frames_per_step": world * B, "batches_in_flight_per_gpu": nfl, "distinct_resident_input_batches_per_gpu": nfl,
                       "sequence_frames_per_rank": len(base),
                       "sharding": "one sequence per GPU, no data-path collective"},
            "stage_ms_per_batch": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "limited_by": "valu issue (byte-wise image work: %s of the kernel's cycles issue VALU "
                                                       "instructions; see profiles/)" % ("%.0f %%" % (100 * valu) if valu else "most"),
                         "kernel": kname, "kernel_stage": roof_stage, "dominant_stage": dom,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": stage_ms[roof_stage],
                         "algorithmic_bytes_per_launch": algo,
                         "valu_issue_utilisation": valu,
                         "valu_source": "SQ_INSTS_VALU x %.2f cycles (the kernel's own mix priced with the measured per-instruction issue costs: "
                                        "profiles/r03_valu_issue.txt, tools/valu_mix.py) / 1024 SIMDs against SQ_BUSY_CYCLES / 32 "
                                        "(tools/pmc_traffic.py)" % valu_cyc},
        }
        if one_at_a_time:
            out["one_batch_at_a_time"] = one_at_a_time
        if exchange:
            out["initial_exchange"] = exchange
        out["render"] = {"frames": len(base), "workers": workers, "seconds": round(t_render, 2)}
        out["parity"] = "bit-exact vs the in-repo CPU oracle; the oracle restates OpenCV 3.4 / Eigen 3.3.7 / PCL 1.9 and is UNPINNED " \
                        "against the real libraries (none can be built here)"
        out["parity_checked_slots"] = parity_checked
        out["parity_check"] = "after the timed loop: %d random slots of each of the %d contexts' last batch (batch %d, %d in flight) " \
                              "against the CPU oracle - keypoint records, descriptors, SearchByProjection match arrays, identical " \
                              "bytes; a mismatch fails the run" % (min(16, B - 1), nfl, B, nfl)
        if config == 5:
            out["algorithmic_bytes_per_frame"] = {k: int(v) for k, v in algo_bytes.items()}
            if not args.no_extras:
                out["aux_per_frame"] = config5_aux(fe.ctx, base, cam)
        if world == 1 and not args.no_extras and config != 5:
            out["other_scenes"] = scene_stage_times(B, local_rank)
            out["other_scenes"]["_about"] = "the same path on the other textures the configs name, one context, one batch at a time: compare with " \
                                            "stage_ms_per_batch / one_batch_at_a_time of this line (room_boxes)"
            fps_hf, ms_hf = host_fed_rate(fe, gray, depth, Tcw, Twc, B, max(4, min(args.steps, 10)), dev)
            out["value_host_fed"] = fps_hf
            out["host_fed"] = {"ms_per_step": ms_hf, "h2d_bytes_per_step": int(gray.nbytes + depth.nbytes),
                               "d2h_bytes_per_step": int(B * fe.ctx.max_kp * (28 + 32 + 4) + 8 * B),
                               "note": "pinned gray + depth H2D on a copy stream overlapped with the previous batch, keypoints / "
                                       "descriptors / matches D2H; link-bound"}
            fe2 = FrontEnd(cam, max_batch=B, device=local_rank)
            fps_sp, ms_sp, thr = host_fed_sparse_rate([fe, fe2], gray, depth, Tcw, Twc, B, max(4, min(args.steps, 10)), dev)
            fe2.ctx.close()
            out["host_fed_sparse_depth"] = {"value": fps_sp, "unit": "frames/s", "ms_per_step": ms_sp,
                                            "h2d_bytes_per_step": int(gray.nbytes + B * fe.ctx.max_kp * 2),
                                            "d2h_bytes_per_step": int(B * fe.ctx.max_kp * (28 + 32 + 4 + 4) + 12 * B),
                                            "gather_threads": thr,
                                            "note": "gray frames only cross the link; the host gathers one raw depth value per keypoint "
                                                    "(sparse-depth glue), two contexts alternate"}
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is reported by the N=1 run only
            out["cpu_baseline"] = cpu_baseline(base, cam, frames=70 if config == 5 else 220, only_orb=config == 5)
        if world == 1 and not args.no_extras and config != 5:
            pipe.close()
            del gray_t, depth_t, gray_ts, depth_ts
            out["full_frontend"] = full_frontend("ICL")
            out["per_frame_latency"] = per_frame_latency("ICL")
    if world > 1 and args.full_frontend:
        # the whole front-end, sharded: every rank on its own device (the rehearsal: all on device 0) with its share of the host's CPUs
        pipe.close()
        del gray_t, depth_t, gray_ts, depth_ts
        torch.cuda.empty_cache()

        def ranks_sync():
            torch.cuda.synchronize()
            dist.barrier()

        ff = full_frontend("ICL", n_frames=int(os.environ.get("DRFE_FF_FRAMES", 512)), reps=int(os.environ.get("DRFE_FF_REPS", 4)), inflight=int(os.environ.get("DRFE_FF_INFLIGHT", 3)),
                           device=local_rank, ranks_on_host=world, sync=ranks_sync)
        ff_el, ff_frames = sharding.reduce_elapsed_and_frames(ff["seconds_timed"], ff["frames_timed"], dev, dist)
        if out is not None:
            out["full_frontend_sharded"] = {"value": ff_frames / ff_el, "unit": "frames/s", "ranks": world, "frames": ff_frames, "seconds_slowest_rank": ff_el,
                                            "rank0": {k: ff[k] for k in ("value", "frames_per_step", "steps_in_flight", "steps_timed", "host_cpus_available", "host_cores_busy_at_this_rate",
                                                                         "step_wall_ms", "frames_handed_back_per_step")},
                                            "sharding": "one sequence per rank, no data-path collective; barrier before and after the timed region, total frames over the slowest rank's time"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        if rehearsal:
            out["rehearsal"] = "%d ranks share ONE device over gloo: control-path check, not a scaling number" % world
        print(json.dumps(out))


if __name__ == "__main__":
    main()
