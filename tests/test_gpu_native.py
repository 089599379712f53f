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
