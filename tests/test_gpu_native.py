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
