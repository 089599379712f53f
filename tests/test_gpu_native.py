"""-m gpu: native (non-Python) callers of the C-ABI.  tests/native/c_caller.c is plain C99 over include/drfe.h;
tests/native/adaptor_caller.cpp drives one frame through include/drfe_adaptor.hpp - the reference's class interfaces
(ORBextractor::operator(), LineSegment::ExtractLineSegment, PlaneDetection) with stand-in container types.  Both must
produce byte-for-byte what the ctypes path produces (which the other test files compare with the oracle)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def _run(exe, *args):
    p = subprocess.run([os.path.join(NATIVE, exe), *map(str, args)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    return p.stdout


def test_c99_caller_matches_ctypes(frames_room, tmp_path):
    from dr_slam_amd import lib
    g = frames_room[0][0]
    (tmp_path / "g.raw").write_bytes(g.tobytes())
    out = _run("c_caller", tmp_path / "g.raw", g.shape[1], g.shape[0], tmp_path / "o.bin")
    assert "c_caller ok" in out
    raw = (tmp_path / "o.bin").read_bytes()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    c = lib.Context()
    try:
        kps, desc = c.orb_extract(g)
    finally:
        c.close()
    assert n == len(kps) > 500
    assert raw[4:4 + 28 * n] == kps.tobytes() and raw[4 + 28 * n:] == desc.tobytes()


def test_cpp_adaptor_matches_ctypes(frames_room, tmp_path):
    from dr_slam_amd import lib, synth
    g, d, _ = frames_room[0]
    cam = synth.TUM3
    (tmp_path / "g.raw").write_bytes(g.tobytes())
    (tmp_path / "d.raw").write_bytes(d.tobytes())
    out = _run("adaptor_caller", tmp_path / "g.raw", tmp_path / "d.raw", g.shape[1], g.shape[0], tmp_path / "o.bin")
    assert "adaptor ok" in out
    raw = (tmp_path / "o.bin").read_bytes()
    nk, nl, npl, d01 = (int(v) for v in np.frombuffer(raw[:16], np.int32))
    c = lib.Context()
    try:
        kps, desc = c.orb_extract(g)
        lines = c.lsd_extract(g)
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        planes = c.planes_ahc(d, K4, float(np.float32(1.0) / np.float32(cam.depth_factor)))
        lvl1 = c.pyramid_level(0, 1)
    finally:
        c.close()
    assert nk == len(kps) and nl == len(lines["lines"]) and npl == len(planes["planes"]) >= 2
    assert d01 == int(np.unpackbits(desc[0] ^ desc[1]).sum())              # ORBmatcher::DescriptorDistance (SWAR)
    o = 16
    assert raw[o:o + 28 * nk] == kps.tobytes(); o += 28 * nk
    assert raw[o:o + 32 * nk] == desc.tobytes(); o += 32 * nk
    assert raw[o:o + 68 * nl] == lines["lines"].tobytes(); o += 68 * nl
    assert raw[o:o + 32 * nl] == lines["desc"].tobytes(); o += 32 * nl
    pc = np.frombuffer(raw[o:o + 48 * npl], np.float64).reshape(npl, 6); o += 48 * npl
    assert np.array_equal(pc[:, :3], planes["planes"]["normal"]) and np.array_equal(pc[:, 3:], planes["planes"]["center"])
    assert raw[o:o + g.size] == planes["seg"].tobytes(); o += g.size
    row = np.frombuffer(raw[o:], np.uint8)
    assert np.array_equal(row, lvl1[19 + 7, 19:-19])                         # mvImagePyramid[1] is the interior ROI


def test_cpp_pipeline_caller_matches_ctypes(frames_room, tmp_path):
    """tests/native/pipeline_caller.cpp: the throughput path without Python - device buffers from the HIP runtime, five batches
    through drfe_pipeline_submit on three contexts, results through drfe_batch_download_async.  The program itself checks that the
    five batches come back identical; here its first batch is compared with the ctypes path on the same frames."""
    import torch
    from dr_slam_amd import synth
    from dr_slam_amd.pipeline import FrontEnd
    cam = synth.TUM3
    (g0, d0, _), (g1, d1, _) = frames_room[0], frames_room[1]
    for name, a in (("g0", g0), ("d0", d0), ("g1", g1), ("d1", d1)):
        (tmp_path / f"{name}.raw").write_bytes(a.tobytes())
    out = _run("pipeline_caller", tmp_path / "g0.raw", tmp_path / "d0.raw", tmp_path / "g1.raw", tmp_path / "d1.raw", cam.w, cam.h,
               tmp_path / "o.bin")
    assert "pipeline_caller ok" in out
    raw = (tmp_path / "o.bin").read_bytes()
    cnt = np.frombuffer(raw[:16], np.int32)
    mc = np.frombuffer(raw[16:32], np.int32)
    fe = FrontEnd(cam, max_batch=4)
    try:
        gray = torch.from_numpy(np.stack([g0, g1, g0, g1])).cuda()
        depth = torch.from_numpy(np.stack([d0, d1, d0, d1]).view(np.int16)).cuda()
        T = np.tile(np.eye(4, dtype=np.float32), (4, 1, 1))
        fe.process(gray, depth, T, T, th=15.0, check_ori=True, stream=0)
        o = 32
        for s in range(2):
            kps, desc = fe.keypoints(s)
            assert cnt[s] == len(kps) > 500
            assert raw[o:o + 28 * len(kps)] == kps.tobytes()
            o += 28 * len(kps)
            assert raw[o:o + 32 * len(kps)] == desc.tobytes()
            o += 32 * len(kps)
        m, n = fe.matches(1)
        assert n == mc[1] > 100
        assert raw[o:o + 4 * cnt[1]] == m[:cnt[1]].astype(np.int32).tobytes()
        assert cnt[2] == cnt[0] and cnt[3] == cnt[1]
    finally:
        fe.ctx.close()


def test_cpp_reference_member_accesses(frames_room, tmp_path):
    """tests/native/members_caller.cpp spells Frame::ComputePlanes (src/Frame.cc:947-979), Frame::ComputePlanes_CAPE (:1096-1121) and
    the LSDmatcher calls as the reference does, on the adaptor's classes: planeDetector.cloud.vertices[j][k],
    .plane_filter.extractedPlanes[i]->normal / ->center, planeDetectionCape.plane_cloud[i] / .plane_params[i].normal / .d,
    std::vector<Vector3d> keylineFunctions, LSDmatcher::SearchByDescriptor / SerachForInitialize / SearchForTriangulation.  What it
    writes must equal what the ctypes path (held to the oracle elsewhere) gives for the same inputs."""
    from dr_slam_amd import lib, synth
    (g0, d, _), (g1, _, _) = frames_room[0], frames_room[1]
    cam = synth.TUM3
    h, w = g0.shape
    for name, a in (("g0", g0), ("g1", g1), ("d", d)):
        (tmp_path / f"{name}.raw").write_bytes(a.tobytes())
    out = _run("members_caller", tmp_path / "g0.raw", tmp_path / "g1.raw", tmp_path / "d.raw", w, h, tmp_path / "o.bin")
    assert "members ok" in out
    raw = (tmp_path / "o.bin").read_bytes()
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    factor = np.float32(1.0) / np.float32(cam.depth_factor)
    c = lib.Context()
    try:
        planes = c.planes_ahc(d, K4, float(factor))
        dm = d.astype(np.float32) * factor
        cape = c.planes_cape(dm, K4, 20, max_merge_dist=50.0)
        l0, l1 = c.lsd_extract(g0), c.lsd_extract(g1)
        has0 = (np.arange(len(l0["desc"])) % 5 != 0).astype(np.uint8)
        has1 = (np.arange(len(l1["desc"])) % 3 == 0).astype(np.uint8)
        n_sd, m_sd = c.lsd_search_by_descriptor(l0["desc"], l1["desc"], has0, mode=0)
        n_in, m_in = c.lsd_search_by_descriptor(l0["desc"], l1["desc"], None, mode=1)
        n_tr, m_tr = c.lsd_search_for_triangulation(l0["desc"], l1["desc"], np.zeros(len(l0["desc"]), np.uint8), has1)
    finally:
        c.close()
    o = 0

    def take(dtype, count):
        nonlocal o
        a = np.frombuffer(raw, dtype, count, o)
        o += a.nbytes
        return a

    # Frame::ComputePlanes: inputCloud of every plane (float casts of readDepthImage's doubles, z > 9 skipped) and d
    assert int(take(np.int32, 1)[0]) == len(planes["planes"]) >= 2
    z = d.astype(np.float64) * np.float64(factor)
    yy, xx = np.mgrid[0:h, 0:w]
    X = np.where(z > 5.0, 0.0, (xx - np.float64(K4[2])) * z / np.float64(K4[0]))
    Y = np.where(z > 5.0, 0.0, (yy - np.float64(K4[3])) * z / np.float64(K4[1]))
    Z = np.where(z > 5.0, 0.0, z)
    for i, p in enumerate(planes["planes"]):
        n, = take(np.int32, 1)
        dd, = take(np.float32, 1)
        pts = take(np.float32, 3 * int(n)).reshape(-1, 3)
        idx = planes["members"][i]
        ref = np.stack([X.ravel()[idx], Y.ravel()[idx], Z.ravel()[idx]], 1).astype(np.float32)
        ref = ref[ref[:, 2] <= np.float32(9.0)]
        assert np.array_equal(pts.view(np.uint32), ref.view(np.uint32)), i
        nrm, ctr = p["normal"], p["center"]
        assert dd == np.float32(-(nrm[0] * ctr[0] + nrm[1] * ctr[1] + nrm[2] * ctr[2]))
    got, = take(np.int32, 1)
    gz, = take(np.float64, 1)
    assert bool(got) == (Z[h // 2, w // 2] != 0) and gz == Z[h // 2, w // 2]
    # Frame::ComputePlanes_CAPE: plane_params and plane_cloud (points of the pixels labelled i + 1, raster order)
    assert int(take(np.int32, 1)[0]) == len(cape["planes"]) >= 2
    zc = dm.astype(np.float64)
    Xc, Yc = (xx - np.float64(K4[2])) * zc / np.float64(K4[0]), (yy - np.float64(K4[3])) * zc / np.float64(K4[1])
    for i, p in enumerate(cape["planes"]):
        rec = take(np.float64, 4)
        assert np.array_equal(rec[:3], p["normal"]) and rec[3] == p["d"]
        n, = take(np.int32, 1)
        pts = take(np.float32, 3 * int(n)).reshape(-1, 3)
        m = cape["seg"] == i + 1
        ref = np.stack([Xc[m], Yc[m], zc[m]], 1).astype(np.float32)
        assert np.array_equal(pts.view(np.uint32), ref.view(np.uint32)), i
    assert take(np.uint8, w * h).tobytes() == cape["seg"].tobytes()
    # lines + LSDmatcher
    nl = take(np.int32, 2)
    assert tuple(nl) == (len(l0["lines"]), len(l1["lines"]))
    assert np.array_equal(take(np.float64, 3 * int(nl[0])).reshape(-1, 3).view(np.uint64), l0["lineF"].view(np.uint64))
    assert int(take(np.int32, 1)[0]) == n_sd and np.array_equal(take(np.int32, int(nl[1])), m_sd)
    n1, nlm = take(np.int32, 2)
    pairs = take(np.int32, 2 * int(nlm)).reshape(-1, 2)
    assert n1 == n_in == nlm and np.array_equal(pairs, np.stack([np.flatnonzero(m_in >= 0), m_in[m_in >= 0]], 1))
    n2, nmp = take(np.int32, 2)
    pairs = take(np.int32, 2 * int(nmp)).reshape(-1, 2)
    assert n2 == n_tr == nmp and np.array_equal(pairs, np.stack([np.flatnonzero(m_tr >= 0), m_tr[m_tr >= 0]], 1))
    assert int(take(np.int32, 1)[0]) == int(np.unpackbits(l0["desc"][0] ^ l1["desc"][0]).sum())
    assert o == len(raw)


def test_cpp_orbmatcher_thirteen_methods(frames_room, tmp_path):
    """tests/native/matcher_caller.cpp: Planar_SLAM::ORBmatcher with the reference's thirteen signatures (include/ORBmatcher.h:41-84)
    on stand-in Frame / KeyFrame / MapPoint types that live on the HOST - include/drfe_adaptor.hpp loads them into slots
    (drfe_frame_load, three slots for eight frame objects: evictions included), flattens the pointer graph, calls the C-ABI and writes
    pointers back / applies Fuse's graph surgery.  Every resulting pointer vector must equal what the ctypes path gives on the slots
    the frames were EXTRACTED in (so drfe_frame_load == extraction residency too); the ctypes matchers are held to the oracle in
    tests/test_gpu_match.py and tests/test_gpu_bow.py."""
    import torch
    from dr_slam_amd import lib, synth, vocabulary as V
    from dr_slam_amd.pipeline import FrontEnd
    cam = synth.TUM3
    rng = np.random.RandomState(77)
    fe = FrontEnd(cam, max_batch=4)
    try:
        c = fe.ctx
        gray = torch.from_numpy(np.stack([f[0] for f in frames_room])).cuda()
        depth = torch.from_numpy(np.stack([f[1] for f in frames_room]).view(np.int16)).cuda()
        Twc64 = np.stack([f[2] for f in frames_room]).astype(np.float64)
        Tcw, Twc = np.linalg.inv(Twc64).astype(np.float32), Twc64.astype(np.float32)
        fe.process(gray, depth, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        kps, desc, ur, z = [], [], [], []
        for s in range(4):
            k, d = fe.keypoints(s)
            u, zz = c.download_stereo(s)
            kps.append(k); desc.append(d); ur.append(u[:len(k)].copy()); z.append(zz[:len(k)].copy())
        N = [len(k) for k in kps]
        scale = c.scale_tables()[0]
        voc = V.make_synthetic(10, 4, seed=5, stop_fraction=0.02)
        voc.upload(c)
        c.bow_transform_batch(voc.L - 2, 4)

        # ---- the map: points A = frame 0's keypoints with depth, B = frame 3's ----
        def unproject(s):
            ok = z[s] > 0
            x = (kps[s]["x"] - np.float32(cam.cx)) * z[s] / np.float32(cam.fx)
            y = (kps[s]["y"] - np.float32(cam.cy)) * z[s] / np.float32(cam.fy)
            pc = np.stack([x, y, z[s], np.ones_like(x)], 1).astype(np.float32)
            return (pc @ Twc[s].T)[:, :3].astype(np.float32), ok

        rows = []
        frame_mp = [np.full(n, -1, np.int32) for n in N]
        for s in (0, 3):
            world, ok = unproject(s)
            for i in np.flatnonzero(ok):
                frame_mp[s][i] = len(rows)
                rows.append((s, i, world[i]))
        M = len(rows)
        mp_world = np.stack([r[2] for r in rows]).astype(np.float32)
        mp_kf = np.array([r[0] for r in rows], np.int32)
        mp_idx = np.array([r[1] for r in rows], np.int32)
        Ow = np.stack([Twc[s][:3, 3] for s in range(4)]).astype(np.float32)
        v = mp_world - Ow[mp_kf]
        dist = np.linalg.norm(v, axis=1).astype(np.float32)
        mp_normal = (v / dist[:, None]).astype(np.float32)
        lvl = np.array([kps[s]["octave"][i] for s, i, _ in rows])
        mp_max = (dist * scale[lvl] * rng.uniform(0.9, 1.3, M)).astype(np.float32)
        mp_min = (mp_max / scale[-1] * rng.uniform(0.5, 1.0, M)).astype(np.float32)
        mp_desc = np.stack([desc[s][i] for s, i, _ in rows])
        mp_nobs = np.where(rng.uniform(size=M) < 0.15, 0, rng.randint(1, 4, M)).astype(np.int32)
        mp_bad = (rng.uniform(size=M) < 0.03).astype(np.int32)
        nA = int((mp_kf == 0).sum())
        for s in (1, 2):                                            # frames 1 and 2 arrive with a few claims on points of A
            pick = rng.uniform(size=N[s]) < 0.1
            frame_mp[s][pick] = rng.randint(0, nA, int(pick.sum()))
        outlier = [(rng.uniform(size=n) < 0.05).astype(np.uint8) for n in N]
        pts = np.zeros(M, lib.FRUSTUM_POINT_DTYPE)
        pts["world"], pts["normal"], pts["min_distance"], pts["max_distance"] = mp_world, mp_normal, mp_min, mp_max
        tracked = c.is_in_frustum(Tcw[1], fe.cam, pts, 0.5)         # what Frame::isInFrustum leaves for frame 1
        tracked["bad"], tracked["obs_positive"], tracked["desc"] = mp_bad, mp_nobs > 0, mp_desc

        T12 = Tcw[0].astype(np.float64) @ Twc[3].astype(np.float64)
        R12, t12, s12 = T12[:3, :3].astype(np.float32), T12[:3, 3].astype(np.float32), np.float32(1.0)
        T12f = Tcw[1].astype(np.float64) @ Twc[2].astype(np.float64)
        K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], np.float64)
        tx = np.array([[0, -T12f[2, 3], T12f[1, 3]], [T12f[2, 3], 0, -T12f[0, 3]], [-T12f[1, 3], T12f[0, 3], 0]])
        F12 = (np.linalg.inv(K).T @ tx @ T12f[:3, :3] @ np.linalg.inv(K)).astype(np.float32)
        Scw8 = Tcw[3].copy()
        Scw12 = Tcw[2].copy(); Scw12[:3, :] *= np.float32(1.05)

        # ---- scene file ----
        blob = [np.array([0x4d415443, 4, M, voc.n_nodes, voc.k, voc.L], np.int32).tobytes(),
                np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, fe.cam.min_x, fe.cam.max_x, fe.cam.min_y, fe.cam.max_y], np.float32).tobytes(),
                np.array([1000, 8, 20, 7], np.int32).tobytes(), np.float32(1.2).tobytes()]
        for s in range(4):
            blob += [np.int32(N[s]).tobytes(), kps[s].tobytes(), desc[s].tobytes(), ur[s].tobytes(), z[s].tobytes(), Tcw[s].tobytes(),
                     Ow[s].tobytes(), frame_mp[s].tobytes(), outlier[s].tobytes()]
        for i in range(M):
            blob += [mp_world[i].tobytes(), mp_normal[i].tobytes(), mp_min[i].tobytes(), mp_max[i].tobytes(), mp_desc[i].tobytes(),
                     np.array([mp_nobs[i], mp_bad[i], mp_kf[i], mp_idx[i], tracked["track_in_view"][i], tracked["level"][i]], np.int32).tobytes(),
                     np.array([tracked["proj_x"][i], tracked["proj_y"][i], tracked["proj_xr"][i], tracked["view_cos"][i]], np.float32).tobytes()]
        blob += [voc.parent.astype(np.int32).tobytes(), voc.desc.tobytes(), voc.weight.astype(np.float64).tobytes(), voc.is_leaf.astype(np.uint8).tobytes(),
                 F12.tobytes(), s12.tobytes(), R12.tobytes(), t12.tobytes(), Scw8.tobytes(), Scw12.tobytes()]
        (tmp_path / "scene.bin").write_bytes(b"".join(blob))
        out = _run("matcher_caller", tmp_path / "scene.bin", tmp_path / "o.bin")
        assert "matcher ok" in out
        raw = np.frombuffer((tmp_path / "o.bin").read_bytes(), np.int32)
        recs, o = [], 0
        while o < len(raw):
            ret, cnt = int(raw[o]), int(raw[o + 1])
            recs.append((ret, raw[o + 2:o + 2 + cnt]))
            o += 2 + cnt
        assert len(recs) == 13

        def frustum(ids):
            p = np.zeros(len(ids), lib.FRUSTUM_POINT_DTYPE)
            ok = ids >= 0
            p[ok] = pts[ids[ok]]
            d = np.zeros((len(ids), 32), np.uint8)
            d[ok] = mp_desc[ids[ok]]
            return p, d

        def decode(res, base, table, n_table):
            """res: the C-ABI's in/out claim array; entries below n_table are new matches (indices into `table`), the others
            the claims the frame arrived with (encoded n_table + i)"""
            outv = base.copy()
            outv[res < 0] = -1
            new = (res >= 0) & (res < n_table)
            outv[new] = table[res[new]]
            return outv

        mp0, mp1, mp2, mp3 = frame_mp
        bad_of = lambda ids: np.where(ids >= 0, mp_bad[np.maximum(ids, 0)], 0).astype(bool)
        obs_of = lambda ids: np.where(ids >= 0, mp_nobs[np.maximum(ids, 0)] > 0, False)

        # 1  SearchByProjection(Cur = 1, Last = 0, 15, false)
        last = np.zeros(N[0], lib.MAPPOINT_DTYPE)
        valid = (mp0 >= 0) & (outlier[0] == 0)
        last["valid"] = valid
        last["obs_positive"] = valid & obs_of(mp0)
        last["world"][valid], last["desc"][valid] = mp_world[mp0[valid]], mp_desc[mp0[valid]]
        init = np.where(mp1 >= 0, N[0] + np.arange(N[1]), -1).astype(np.int32)
        n, res = c.search_by_projection_last(1, 0, Tcw[1], Tcw[0], fe.cam, last, N[1], 15.0, False, True, cur_mp=init, cur_obs=obs_of(mp1).astype(np.uint8))
        assert recs[0][0] == n > 100 and np.array_equal(recs[0][1], decode(res, mp1, mp0, N[0]))
        # 2  MatchORBPoints(Cur = 2, Last = 0)
        init = np.where(mp2 >= 0, N[0] + np.arange(N[2]), -1).astype(np.int32)
        n, res = c.match_orb_points(2, 0, np.where(mp0 >= 0, np.arange(N[0]), -1), outlier[0], N[2], cur_mp=init)
        assert recs[1][0] == n > 50 and np.array_equal(recs[1][1], decode(res, mp2, mp0, N[0]))
        # 3  SearchByProjection(F = 1, local map, 3), nnratio 0.8
        init = np.where(mp1 >= 0, M + np.arange(N[1]), -1).astype(np.int32)
        n, res = c.search_by_projection_map(1, tracked, N[1], 3.0, 0.8, frame_mp=init, claim_obs=obs_of(mp1).astype(np.uint8))
        assert recs[2][0] == n > 100 and np.array_equal(recs[2][1], decode(res, mp1, np.arange(M, dtype=np.int32), M))
        # 4  SearchByBoW(KF 0, F 1), ORBmatcher(0.7, true)
        n, m = c.search_by_bow(0, 1, np.where((mp0 >= 0) & ~bad_of(mp0), np.arange(N[0]), -1), N[1], 0.7, True)
        assert recs[3][0] == n > 50 and np.array_equal(recs[3][1], np.where(m >= 0, mp0[np.maximum(m, 0)], -1))
        # 5  SearchByBoW(KF 0, KF 3), ORBmatcher(0.75, true)
        n, m2 = c.search_by_bow_kf(0, 3, np.where((mp0 >= 0) & ~bad_of(mp0), np.arange(N[0]), -1),
                                   np.where((mp3 >= 0) & ~bad_of(mp3), np.arange(N[3]), -1), 0.75, True)
        m12 = np.full(N[0], -1, np.int32)
        m12[m2[m2 >= 0]] = mp3[np.flatnonzero(m2 >= 0)]
        assert recs[4][0] == n > 20 and np.array_equal(recs[4][1], m12)
        # 6  SearchForTriangulation(KF 1, KF 2, F12, pairs, false), ORBmatcher(0.6, false)
        n, mt = c.search_for_triangulation(1, 2, np.where(mp1 >= 0, np.arange(N[1]), -1), np.where(mp2 >= 0, np.arange(N[2]), -1), F12, Ow[1], Tcw[2],
                                           fe.cam, False, False)
        pairs = np.stack([np.flatnonzero(mt >= 0), mt[mt >= 0]], 1).astype(np.int32).reshape(-1)
        assert recs[5][0] == n and np.array_equal(recs[5][1], pairs)
        # 7  SearchBySim3(KF 0, KF 3, vpMatches12 of call 5 with every third entry cleared, 1.0, R12, t12, 7.5)
        start = m12.copy(); start[::3] = -1
        skip1 = (start >= 0) | (mp0 < 0) | bad_of(mp0)
        skip2 = (mp3 < 0) | bad_of(mp3)
        held = start[start >= 0]
        skip2[mp_idx[held[mp_kf[held] == 3]]] = True
        p1, d1 = frustum(np.where(skip1, -1, mp0)); p2, d2 = frustum(np.where(skip2, -1, mp3))
        n, ms = c.search_by_sim3(0, 3, Tcw[0], Tcw[3], 1.0, R12, t12, p1, d1, skip1.astype(np.uint8), p2, d2, skip2.astype(np.uint8), 7.5)
        want = start.copy(); want[ms >= 0] = mp3[ms[ms >= 0]]
        assert recs[6][0] == n > 20 and np.array_equal(recs[6][1], want)
        # 8  SearchByProjection(KF 3, Scw, points of KF 0 twice each, vpMatched, 10)
        lst = np.repeat(mp0[mp0 >= 0], 2)
        vm = np.full(N[3], -1, np.int32); vm[::5] = mp3[::5]
        p, d = frustum(lst)
        n, new = c.search_by_projection_kf(3, Scw8, p, d, bad_of(lst).astype(np.uint8), (vm >= 0).astype(np.uint8), 10.0)
        want = vm.copy(); want[new >= 0] = lst[new[new >= 0]]
        assert recs[7][0] == n > 100 and np.array_equal(recs[7][1], want)
        # 9  SearchByProjection(Cur = 2, KF 0, sAlreadyFound, 10, 100) (relocalisation)
        found = set(int(v) for v in mp2[mp2 >= 0]) | set(int(v) for v in mp0[::7] if v >= 0)
        skip = (mp0 < 0) | bad_of(mp0) | np.array([int(v) in found for v in mp0])
        p, d = frustum(np.where(skip, -1, mp0))
        n, new = c.search_by_projection_reloc(2, Tcw[2], p, d, kps[0]["angle"], skip.astype(np.uint8), (mp2 >= 0).astype(np.uint8), 10.0, 100, True)
        want = mp2.copy(); want[new >= 0] = mp0[new[new >= 0]]
        assert recs[8][0] == n > 100 and np.array_equal(recs[8][1], want)
        # 10  SearchForInitialization(F 0, F 1, prev, matches, 100), ORBmatcher(0.9, true)
        prev = np.stack([kps[0]["x"], kps[0]["y"]], 1).astype(np.float32)
        n, mi, prev2 = c.search_for_initialization(0, 1, prev, 100, 0.9, True)
        assert recs[9][0] == n > 50 and np.array_equal(recs[9][1][:N[0]], mi) and np.array_equal(recs[9][1][N[0]:], prev2.reshape(-1).view(np.int32))
        # 11  Fuse(KF 1, points of KF 0 with NULLs and repeats, 3.0): search through ctypes, the surgery replayed here
        lst = mp0.copy()
        for i in range(0, len(lst) - 1, 11):
            lst[i + 1] = lst[i]
        bad, nobs, repl = mp_bad.astype(bool).copy(), mp_nobs.copy(), np.full(M, -1, np.int32)
        in_kf1 = np.zeros(M, bool)
        p, d = frustum(lst)
        bi, bd = c.fuse_search(1, Tcw[1], p, d, ((lst < 0) | bad_of(lst)).astype(np.uint8), 3.0)
        kf1 = mp1.copy()
        fused = 0
        for i, pid in enumerate(lst):
            if pid < 0 or bad[pid] or in_kf1[pid] or bi[i] < 0 or bd[i] > 50:
                continue
            inkf = kf1[bi[i]]
            if inkf >= 0:
                if not bad[inkf] and inkf != pid:
                    if nobs[inkf] > nobs[pid]:
                        bad[pid], repl[pid] = True, inkf
                    else:
                        bad[inkf], repl[inkf] = True, pid
            else:
                in_kf1[pid] = True; nobs[pid] += 1; kf1[bi[i]] = pid
            fused += 1
        assert recs[10][0] == fused > 100
        assert np.array_equal(recs[10][1][:N[1]], kf1)
        assert np.array_equal(recs[10][1][N[1]:].reshape(M, 3), np.stack([bad.astype(np.int32), repl, nobs], 1))
        assert repl.max() >= 0 and (nobs != mp_nobs).any()                      # both branches of the surgery ran
        # 12  Fuse(KF 2, Scw, points of KF 3, 4.0, vpReplacePoint)
        lst = mp3[mp3 >= 0]
        already = set(int(v) for v in mp2[mp2 >= 0] if not bad[v])
        skip = np.array([bool(bad[v]) or int(v) in already for v in lst])
        p, d = frustum(lst)
        bi, bd = c.fuse_search_sim3(2, Scw12, p, d, skip.astype(np.uint8), 4.0)
        kf2, rp, fused = mp2.copy(), np.full(len(lst), -1, np.int32), 0
        for i, pid in enumerate(lst):
            if skip[i] or bi[i] < 0 or bd[i] > 50:
                continue
            inkf = kf2[bi[i]]
            if inkf >= 0:
                if not bad[inkf]:
                    rp[i] = inkf
            else:
                kf2[bi[i]] = pid
            fused += 1
        assert recs[11][0] == fused > 20
        assert np.array_equal(recs[11][1][:len(lst)], rp) and np.array_equal(recs[11][1][len(lst):], kf2)
        # 13  DescriptorDistance(cv::Mat, cv::Mat); eight frame objects went through three slots
        assert recs[12][0] == int(np.unpackbits(mp_desc[0] ^ mp_desc[-1]).sum()) and recs[12][1][0] >= 8
    finally:
        fe.ctx.close()


def test_cpp_lsdmatcher_ten_methods(frames_room, tmp_path):
    """tests/native/linematcher_caller.cpp: Planar_SLAM::LSDmatcher with the reference's ten signatures (include/LSDmatcher.h:21-36) on
    stand-in Frame / KeyFrame / MapLine types - include/drfe_adaptor.hpp flattens the pointer graph, calls the C-ABI, writes MapLine*
    back and applies Fuse's Replace / AddObservation / AddMapLine in the reference's order (the stand-in MapLine::Replace moves
    observations between keyframes as src/MapLine.cpp:178-214 does, so later lines of the loop meet what earlier ones left).  Every
    resulting pointer vector must equal what the ctypes path gives on the flattened records + a replay of the surgery here; the
    ctypes searches are held to the oracle in tests/test_gpu_lines.py.  Call 12: LineSegment::ExtractLineSegment from a fresh thread
    on a default-constructed object (the reference's mpLineSegment is never initialised, include/Frame.h:157)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import line_scenarios as LS
    from dr_slam_amd import lib
    rng = np.random.RandomState(4242)
    KL, MLD, TLD, FLD = lib.KEYLINE_DTYPE, lib.MAPLINE_DTYPE, lib.TRACKED_LINE_DTYPE, lib.FRUSTUM_LINE_DTYPE
    ML_DT = np.dtype([("world", "<f8", (6,)), ("normal", "<f8", (3,)), ("min", "<f4"), ("max", "<f4"), ("desc", "u1", (32,)), ("nobs", "<i4"),
                      ("bad", "<i4"), ("okf", "<i4"), ("oidx", "<i4"), ("inview", "<i4"), ("level", "<i4"), ("x1", "<f4"), ("y1", "<f4"),
                      ("x2", "<f4"), ("y2", "<f4"), ("vcos", "<f4")])
    assert ML_DT.itemsize == 48 + 24 + 8 + 32 + 24 + 20
    tabs = []

    def new_lines(n):
        t = np.zeros(n, ML_DT)
        t["okf"] = -1
        base = sum(len(x) for x in tabs)
        tabs.append(t)
        return base, t

    # ---- scenario A: SearchByProjection(Cur, Last) and (F, local map lines) ----
    n_cur, n_last = 40, 48
    scA = LS.make(1, KL, MLD, TLD, n_cur=n_cur, n_last=n_last)
    bA, A = new_lines(n_last)
    A["world"], A["desc"] = scA["last"]["world"], scA["last"]["desc"]
    A["nobs"] = np.where(scA["last"]["obs_positive"] != 0, 2, 0)
    A["bad"] = rng.uniform(size=n_last) < 0.06
    A["inview"] = rng.uniform(size=n_last) < 0.9
    A["level"] = scA["tracked"]["level"]
    for k in ("x1", "y1", "x2", "y2"):
        A[k] = scA["tracked"][k]
    A["vcos"] = scA["tracked"]["view_cos"]
    last_ml = np.where(rng.uniform(size=n_last) < 0.08, -1, bA + np.arange(n_last)).astype(np.int32)
    last_out = (rng.uniform(size=n_last) < 0.06).astype(np.uint8)
    pre_j = np.flatnonzero(scA["cur_ml"] >= 0)
    bP, P = new_lines(len(pre_j))
    P["nobs"] = scA["cur_obs"][pre_j]
    cur_ml = np.full(n_cur, -1, np.int32)
    cur_ml[pre_j] = bP + np.arange(len(pre_j))
    last_kl = np.zeros(n_last, KL)
    last_kl["octave"] = scA["last"]["octave"]
    local = (bA + np.arange(n_last)).astype(np.int32)
    local[::9] = -1

    # ---- scenario B: the descriptor matchers ----
    n_a, n_b = 40, 37
    desc_a = rng.randint(0, 256, (n_a, 32)).astype(np.uint8)
    perm = rng.permutation(n_a)
    desc_b = rng.randint(0, 256, (n_b, 32)).astype(np.uint8)
    for t in range(30):
        desc_b[t] = LS._flip(rng, desc_a[perm[t]], int(rng.choice([0, 5, 20, 45, 80])))
    bBa, Ba = new_lines(n_a)
    bBb, Bb = new_lines(n_b)
    a_ml = np.where(rng.uniform(size=n_a) < 0.8, bBa + np.arange(n_a), -1).astype(np.int32)
    b_ml = np.where(rng.uniform(size=n_b) < 0.8, bBb + np.arange(n_b), -1).astype(np.int32)

    # ---- scenario C: Fuse x2 and SearchByProjection(KF, Scw) on one keyframe ----
    n_kf, n_c = 40, 240
    scC, linesC, rngC = LS.sim3_line_scene(2, n_kf, n_c, KL, MLD, TLD, FLD)
    bC, Cl = new_lines(n_c)
    Cl["world"], Cl["normal"], Cl["min"], Cl["max"] = linesC["world"], linesC["normal"], linesC["min_distance"], linesC["max_distance"]
    Cl["desc"] = scC["last"]["desc"]
    Cl["nobs"] = rng.randint(0, 4, n_c)
    Cl["bad"] = rng.uniform(size=n_c) < 0.08
    bK, Kp = new_lines(n_kf)
    Kp["nobs"] = rng.randint(0, 4, n_kf)
    Kp["bad"] = rng.uniform(size=n_kf) < 0.1
    kf4_ml = np.full(n_kf, -1, np.int32)
    for k in range(n_kf):
        if rng.uniform() < 0.35:
            kf4_ml[k] = bK + k
            Kp["okf"][k], Kp["oidx"][k] = 4, k
    for k, i in ((3, 5), (17, 21), (31, 140)):                       # three lines of the candidate list already sit in the keyframe
        if kf4_ml[k] >= 0:
            Kp["okf"][kf4_ml[k] - bK] = -1
        kf4_ml[k] = bC + i
        Cl["okf"][i], Cl["oidx"][i] = 4, k
    fuse_lst = (bC + np.arange(n_c)).astype(np.int32)
    for i in range(0, n_c - 1, 11):
        fuse_lst[i + 1] = fuse_lst[i]
    fuse_lst[5::13] = -1
    fuse2_lst = (bC + np.arange(n_c)).astype(np.int32)
    fuse2_lst[7::17] = -1
    proj_lst = np.concatenate([bC + np.arange(n_c), bC + np.arange(n_c)]).astype(np.int32)
    proj_lst[::23] = -1
    TcwC = scC["Tcw_cur"].astype(np.float32)
    Scw9 = TcwC.copy(); Scw9[:3, :] *= np.float32(0.83)
    Scw10 = TcwC.copy(); Scw10[:3, :] *= np.float32(1.7)

    # ---- scenario D: SearchBySim3 ----
    K = LS.keyframe_pair_for_sim3(1, KL, FLD)
    n_d = K["n"]
    bD1, D1 = new_lines(n_d)
    bD2, D2 = new_lines(n_d)
    for D, ln, ds, kfi in ((D1, K["lines1"], K["descs1"], 5), (D2, K["lines2"], K["descs2"], 6)):
        D["world"], D["normal"], D["min"], D["max"], D["desc"] = ln["world"], ln["normal"], ln["min_distance"], ln["max_distance"], ds
        D["nobs"] = 1
        D["okf"], D["oidx"] = kfi, np.arange(n_d)
    odd = (np.arange(n_d) % 2) == 1
    D1["bad"] = (K["skip1"] != 0) & odd
    D2["bad"] = (K["skip2"] != 0) & odd
    kf5_ml = np.where((K["skip1"] != 0) & ~odd, -1, bD1 + np.arange(n_d)).astype(np.int32)
    kf6_ml = np.where((K["skip2"] != 0) & ~odd, -1, bD2 + np.arange(n_d)).astype(np.int32)
    D1["okf"][kf5_ml < 0] = -1
    D2["okf"][kf6_ml < 0] = -1
    inv = np.argsort(K["perm"])                                      # key line of KF2 that shows world line i
    sim3_init = np.full(n_d, -1, np.int32)
    for i1 in range(0, n_d, 6):
        if kf6_ml[inv[i1]] >= 0:
            sim3_init[i1] = bD2 + inv[i1]

    ML = np.concatenate(tabs)
    nML = len(ML)
    img = frames_room[0][0]
    h, w = img.shape

    # ---- scene file ----
    zero16 = np.eye(4, dtype=np.float32)
    frames = [(scA["cur"], scA["cur_desc"], scA["Tcw_cur"], cur_ml, np.zeros(n_cur, np.uint8)),
              (last_kl, np.zeros((n_last, 32), np.uint8), scA["Tcw_last"], last_ml, last_out),
              (np.zeros(n_a, KL), desc_a, zero16, a_ml, np.zeros(n_a, np.uint8)),
              (np.zeros(n_b, KL), desc_b, zero16, b_ml, np.zeros(n_b, np.uint8)),
              (scC["cur"], scC["cur_desc"], TcwC, kf4_ml, np.zeros(n_kf, np.uint8)),
              (K["kl1"], K["kd1"], K["T1w"], kf5_ml, np.zeros(n_d, np.uint8)),
              (K["kl2"], K["kd2"], K["T2w"], kf6_ml, np.zeros(n_d, np.uint8))]
    ints = lambda v: np.int32(len(v)).tobytes() + np.ascontiguousarray(v, np.int32).tobytes()
    blob = [np.array([0x4c494e45, nML, len(frames)], np.int32).tobytes(),
            np.array([LS.CAM[k] for k in ("fx", "fy", "cx", "cy", "bf", "min_x", "max_x", "min_y", "max_y")], np.float32).tobytes(),
            np.array([1000, 8, 20, 7], np.int32).tobytes(), np.float32(1.2).tobytes(), ML.tobytes()]
    for kl, ds, T, mlid, outl in frames:
        blob += [np.int32(len(kl)).tobytes(), np.ascontiguousarray(kl, KL).tobytes(), np.ascontiguousarray(ds, np.uint8).tobytes(),
                 np.ascontiguousarray(T, np.float32).tobytes(), np.ascontiguousarray(mlid, np.int32).tobytes(), outl.tobytes()]
    blob += [ints(local), np.float32(15.0).tobytes(), np.float32(2.0).tobytes(), ints(fuse_lst), np.float32(12.0).tobytes(), Scw9.tobytes(), Scw10.tobytes(), ints(fuse2_lst),
             ints(proj_lst)]
    # vpMatched of call 9 depends on what calls 7 and 8 leave in the keyframe: computed by the replay below, so the file is
    # written after it

    ctx = lib.Context(max_batch=1)
    try:
        cam = lib.Camera(**LS.CAM)
        bad = ML["bad"].astype(bool).copy()
        nobs = ML["nobs"].copy()
        repl = np.full(nML, -1, np.int32)
        obs = [dict() for _ in range(nML)]
        for i in range(nML):
            if ML["okf"][i] >= 0:
                obs[i][int(ML["okf"][i])] = int(ML["oidx"][i])
        kfml = {4: kf4_ml.copy(), 5: kf5_ml.copy(), 6: kf6_ml.copy()}

        def add_observation(p, kf, idx):
            if kf in obs[p]:
                return
            obs[p][kf] = idx
            nobs[p] += 1

        def replace(a, b):                                           # a->Replace(b), src/MapLine.cpp:178-214
            if a == b:
                return
            o, obs[a] = obs[a], dict()
            bad[a], repl[a] = True, b
            for kf in sorted(o):
                if kf not in obs[b]:
                    kfml[kf][o[kf]] = b
                    add_observation(b, kf, o[kf])
                else:
                    kfml[kf][o[kf]] = -1

        def frustum(lst):
            f = np.zeros(len(lst), FLD)
            d = np.zeros((len(lst), 32), np.uint8)
            ok = lst >= 0
            src = ML[lst[ok]]
            f["world"][ok], f["normal"][ok], f["min_distance"][ok], f["max_distance"][ok] = src["world"], src["normal"], src["min"], src["max"]
            d[ok] = src["desc"]
            return f, d

        want = []
        # 1
        rec = scA["last"].copy()
        rec["valid"] = (last_ml >= 0) & ~bad[np.maximum(last_ml, 0)] & (last_out == 0)
        init = np.where(cur_ml >= 0, n_last + np.arange(n_cur), -1).astype(np.int32)
        n, res = ctx.lsd_search_by_projection_last(scA["Tcw_cur"], scA["Tcw_last"], cam, rec, scA["cur"], scA["cur_desc"], 15.0, False, 0.9, init,
                                                   scA["cur_obs"])
        w1 = cur_ml.copy()
        new = (res >= 0) & (res < n_last)
        w1[new] = last_ml[res[new]]
        assert n > 5 and new.sum() > 5
        want.append((n, w1))
        # 2
        rec = scA["tracked"].copy()
        rec["in_view"] = (local >= 0) & ~bad[bA:bA + n_last] & (A["inview"] != 0)
        rec["obs_positive"] = A["nobs"] > 0
        init = np.where(cur_ml >= 0, n_last + np.arange(n_cur), -1).astype(np.int32)
        n, res = ctx.lsd_search_by_projection_map(rec, scA["cur"], scA["cur_desc"], 2.0, 0.9, init, scA["cur_obs"])
        w2 = cur_ml.copy()
        new = (res >= 0) & (res < n_last)
        w2[new] = local[res[new]]
        assert n > 5
        want.append((n, w2))
        # 3 - 6
        n, m = ctx.lsd_search_by_descriptor(desc_a, desc_b, (a_ml >= 0).astype(np.uint8), mode=0)
        assert n >= 5
        want.append((n, np.where(m >= 0, a_ml[np.maximum(m, 0)], -1)))
        n, m = ctx.lsd_search_by_descriptor(desc_a, desc_b, (b_ml >= 0).astype(np.uint8), mode=1)
        assert n >= 5
        want.append((n, np.where(m >= 0, b_ml[np.maximum(m, 0)], -1)))
        n, m = ctx.lsd_search_by_descriptor(desc_a, desc_b, None, mode=1)
        assert n >= 5
        want.append((n, np.stack([np.flatnonzero(m >= 0), m[m >= 0]], 1).reshape(-1)))
        n, m = ctx.lsd_search_for_triangulation(desc_a, desc_b, (a_ml >= 0).astype(np.uint8), (b_ml >= 0).astype(np.uint8))
        want.append((n, np.stack([np.flatnonzero(m >= 0), m[m >= 0]], 1).reshape(-1)))
        # 7  Fuse(KF 4, lines with NULLs and repeats, 12.0)
        kl4, kd4 = scC["cur"], scC["cur_desc"]
        f, d = frustum(fuse_lst)
        skip = (fuse_lst < 0) | bad[np.maximum(fuse_lst, 0)]
        bi, bd = ctx.lsd_fuse_search(TcwC, cam, f, d, skip.astype(np.uint8), kl4, kd4, 12.0)
        fused = 0
        for i, p in enumerate(fuse_lst):
            if p < 0 or bad[p] or bi[i] < 0 or bd[i] > 50:
                continue
            inkf = kfml[4][bi[i]]
            if inkf >= 0:
                if not bad[inkf]:
                    if nobs[inkf] > nobs[p]:
                        replace(p, inkf)
                    else:
                        replace(inkf, p)
            else:
                add_observation(p, 4, int(bi[i]))
                kfml[4][bi[i]] = p
            fused += 1
        assert fused > 8 and (repl >= 0).sum() >= 2 and (nobs != ML["nobs"]).any()
        want.append((fused, np.concatenate([kfml[4], np.stack([bad.astype(np.int32), repl, nobs], 1).reshape(-1)])))
        # 8  Fuse(KF 4, Scw, lines, 4.0, vpReplaceLine) on that state
        f, d = frustum(fuse2_lst)
        already = set(int(v) for v in kfml[4] if v >= 0 and not bad[v])
        skip = np.array([(p < 0) or bool(bad[p]) or (int(p) in already) for p in fuse2_lst])
        bi, bd = ctx.lsd_fuse_search_sim3(Scw9, cam, f, d, skip.astype(np.uint8), kl4, kd4, 4.0)
        rp, fused = np.full(len(fuse2_lst), -1, np.int32), 0
        for i, p in enumerate(fuse2_lst):
            if skip[i] or bi[i] < 0 or bd[i] > 50:
                continue
            inkf = kfml[4][bi[i]]
            if inkf >= 0:
                if not bad[inkf]:
                    rp[i] = inkf
            else:
                add_observation(p, 4, int(bi[i]))
                kfml[4][bi[i]] = p
            fused += 1
        assert fused > 3
        want.append((fused, np.concatenate([rp, kfml[4]])))
        # 9  SearchByProjection(KF 4, Scw, lines twice, vpMatched, 10)
        vm = np.where(np.arange(n_kf) % 3 == 0, kfml[4], -1).astype(np.int32)
        f, d = frustum(proj_lst)
        found = set(int(v) for v in vm if v >= 0)
        skip = np.array([(p < 0) or bool(bad[p]) or (int(p) in found) for p in proj_lst])
        n, new = ctx.lsd_search_by_projection_kf(Scw10, cam, f, d, skip.astype(np.uint8), kl4, kd4, (vm >= 0).astype(np.uint8), 10)
        w9 = vm.copy()
        w9[new >= 0] = proj_lst[new[new >= 0]]
        assert n > 5
        want.append((n, w9))
        # 10  SearchBySim3(KF 5, KF 6, vpMatches12, s12, R12, t12, 7.5)
        skip1 = (kf5_ml < 0) | bad[np.maximum(kf5_ml, 0)] | (sim3_init >= 0)
        skip2 = (kf6_ml < 0) | bad[np.maximum(kf6_ml, 0)]
        for v in sim3_init[sim3_init >= 0]:
            skip2[obs[v][6]] = True
        n, m = ctx.lsd_search_by_sim3(cam, K["T1w"], K["T2w"], K["s12"], K["R12"], K["t12"], K["lines1"], K["descs1"], skip1.astype(np.uint8), K["kl1"],
                                      K["kd1"], K["lines2"], K["descs2"], skip2.astype(np.uint8), K["kl2"], K["kd2"], 7.5)
        w10 = sim3_init.copy()
        w10[m >= 0] = kf6_ml[m[m >= 0]]
        assert n > 5 and (sim3_init >= 0).sum() >= 3
        want.append((n, w10))
        # 11
        want.append((int(np.unpackbits(ML["desc"][0] ^ ML["desc"][-1]).sum()), np.zeros(0, np.int32)))
        # 12
        lines = ctx.lsd_extract(img)
        want.append((len(lines["lines"]), np.concatenate([lines["desc"].reshape(-1).view(np.int32),
                                                          np.stack([lines["lines"]["start_point_x"], lines["lines"]["end_point_y"]], 1).reshape(-1).view(np.int32)])))
        assert len(lines["lines"]) > 5
    finally:
        ctx.close()

    blob += [ints(vm), ints(sim3_init), np.float32(K["s12"]).tobytes(), np.ascontiguousarray(K["R12"], np.float32).tobytes(),
             np.ascontiguousarray(K["t12"], np.float32).tobytes(), np.array([w, h], np.int32).tobytes(), img.tobytes()]
    (tmp_path / "scene.bin").write_bytes(b"".join(blob))
    out = _run("linematcher_caller", tmp_path / "scene.bin", tmp_path / "o.bin")
    assert "linematcher ok" in out
    raw = np.frombuffer((tmp_path / "o.bin").read_bytes(), np.int32)
    recs, o = [], 0
    while o < len(raw):
        ret, cnt = int(raw[o]), int(raw[o + 1])
        recs.append((ret, raw[o + 2:o + 2 + cnt]))
        o += 2 + cnt
    assert len(recs) == len(want) == 12
    for k, ((ret, vec), (wret, wvec)) in enumerate(zip(recs, want)):
        assert ret == wret, (k + 1, ret, wret)
        assert np.array_equal(vec, np.asarray(wvec, np.int32)), k + 1
