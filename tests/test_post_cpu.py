"""CPU tests of the plane post-processing (Frame::ComputePlanes after the extractor):
the oracle (oracle/post_oracle.cpp, a restatement of PCL 1.9.1 - parity unpinned) against independent float64 / numpy
definitions, and the product's HOST entry points (drfe_plane_voxel_grid / drfe_plane_refit: no device work) against the
oracle bit for bit."""
import numpy as np
import pytest


def _plane_points(seed, n=4000, noise=0.004):
    rng = np.random.default_rng(seed)
    nrm = np.array([0.2, -0.5, -0.84])
    nrm /= np.linalg.norm(nrm)
    d = 1.7
    a = np.cross(nrm, [1.0, 0, 0]); a /= np.linalg.norm(a)
    b = np.cross(nrm, a)
    uv = rng.uniform(-1.2, 1.2, (n, 2))
    p = -d * nrm + uv[:, :1] * a + uv[:, 1:] * b + rng.normal(0, noise, (n, 1)) * nrm
    return p.astype(np.float32), nrm, d


def test_mt19937_known_answers(oracle_mod):
    """The sample-consensus RNG: the C++11 known answer (10000th output of seed 5489) and numpy's MT19937 for the seed
    PCL uses (12345); uniform_int<>(0, INT_MAX) over it is the output >> 1."""
    L = oracle_mod.lib()
    import ctypes as C
    L.orc_post_mt19937.restype = C.c_uint32
    L.orc_post_mt19937.argtypes = [C.c_uint32, C.c_int]
    assert L.orc_post_mt19937(5489, 9999) == 4123659995
    ref = np.random.RandomState(12345).randint(0, 2**32, 50, dtype=np.uint64)
    assert [L.orc_post_mt19937(12345, i) for i in range(50)] == [int(v) for v in ref]


def test_voxel_grid_matches_float64_definition(oracle_mod):
    pts, _, _ = _plane_points(1)
    vox = oracle_mod.post_voxel_grid(pts, 0.05)
    inv = np.float32(1.0) / np.float32(0.05)
    ijk = np.floor(pts * inv).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    key = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    uk = np.unique(key)
    assert len(vox) == len(uk) and 100 < len(vox) < len(pts)
    ref = np.stack([pts[key == k].astype(np.float64).mean(0) for k in uk])
    assert np.abs(vox - ref).max() < 2e-6               # leaves in ascending index order, float32 sums
    # ragged inputs
    assert len(oracle_mod.post_voxel_grid(np.zeros((0, 3), np.float32))) == 0
    one = oracle_mod.post_voxel_grid(pts[:1])
    assert one.shape == (1, 3) and np.array_equal(one[0], pts[0])


def test_refit_recovers_the_plane(oracle_mod):
    pts, nrm, d = _plane_points(2)
    vox = oracle_mod.post_voxel_grid(pts)
    coef0 = np.array([*nrm, d], np.float32) + np.array([0.01, -0.01, 0.0, 0.004], np.float32)
    ok, coef = oracle_mod.post_refit(coef0, vox, 0.05)
    assert ok
    assert abs(np.linalg.norm(coef[:3]) - 1) < 1e-5
    assert np.abs(coef[:3] - nrm).max() < 5e-3 and abs(coef[3] - d) < 5e-3      # least squares over the inliers
    # sign stays on the side of the extractor's d
    ok, flipped = oracle_mod.post_refit(-coef0, vox, 0.05)
    assert ok and flipped[3] < 0 and np.abs(flipped + coef).max() < 1e-6
    # one voxel point farther than Plane.DistanceThreshold rejects the plane outright (src/Frame.cc:1238-1242)
    bad = np.vstack([vox, (vox[0] + 0.2 * nrm.astype(np.float32))[None]])
    assert oracle_mod.post_refit(coef0, bad, 0.05)[0] is False
    assert oracle_mod.post_refit(coef0, vox[:2], 0.05)[0] is False              # fewer than three points


def test_host_entry_points_equal_the_oracle(oracle_mod):
    """drfe_plane_voxel_grid / drfe_plane_refit are host code behind the C-ABI (std::mt19937 in the product, a hand-written
    twister in the oracle; structurally different plane-fit code): bit-equal outputs on seeded clouds."""
    from dr_slam_amd import lib
    for seed in range(6):
        pts, nrm, d = _plane_points(10 + seed, n=1500 + 700 * seed, noise=0.002 * (1 + seed))
        a, b = lib.plane_voxel_grid(pts), oracle_mod.post_voxel_grid(pts)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        coef0 = np.array([*nrm, d], np.float32)
        for th in (0.05, 0.10, 0.012):
            (va, ca), (vb, cb) = lib.plane_refit(coef0, a, th), oracle_mod.post_refit(coef0, b, th)
            assert va == vb and np.array_equal(ca.view(np.uint32), cb.view(np.uint32)), (seed, th)
    assert len(lib.plane_voxel_grid(np.zeros((0, 3), np.float32))) == 0


def test_surface_normals_on_a_tilted_plane(oracle_mod):
    """pcl::IntegralImageNormalEstimation (AVERAGE_3D_GRADIENT) restated: on a noise-free tilted plane every defined normal
    is the plane normal turned towards the camera; the 10-point border, the points next to a depth discontinuity and the
    region beyond Point.MaxDistance are NaN / point at zero depth."""
    from dr_slam_amd import synth
    cam = synth.TUM3
    h, w = cam.h, cam.w
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    n = np.array([0.15, 0.35, -0.92]); n /= np.linalg.norm(n)
    ray = np.stack([(u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u)], -1)
    z = (-2.0 / (ray @ n)).astype(np.float32)           # plane n.p + 2 = 0
    z[:, 400:] += 1.0                                    # a 1 m step: depth discontinuity
    K4 = (cam.fx, cam.fy, cam.cx, cam.cy)
    cloud, nrm = oracle_mod.post_surface_normals(z, K4, 9.0)
    H, W = nrm.shape[:2]
    assert (H, W) == (160, 214)
    ok = np.isfinite(nrm[..., 0])
    assert not ok[:10].any() and not ok[-10:].any() and not ok[:, :10].any() and not ok[:, -10:].any()
    assert ok[10:-10, 10:120].all()
    good = nrm[ok]
    assert np.abs(np.linalg.norm(good, axis=1) - 1).max() < 1e-6
    inner = nrm[20:-20, 20:110].reshape(-1, 3)
    assert np.abs(inner - n.astype(np.float32)).max() < 2e-3     # towards the camera: n.z < 0
    step_col = 400 // 3
    assert not ok[:, step_col - 1:step_col + 2].any()            # distance map <= 2 next to the jump
    # beyond Point.MaxDistance the cloud is (0, 0, 0)
    cloud2, nrm2 = oracle_mod.post_surface_normals(z, K4, 2.5)
    assert (cloud2[z[::3, ::3] > 2.5] == 0).all()
    recs = oracle_mod.post_surface_normal_records(cloud, nrm)
    assert len(recs[0]) == 80 * 107 and recs[2][0] == 3 and recs[3][0] == 3


GOLD_POST = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "post_room_320x240.npz")


def test_golden_post(oracle_mod):
    """Committed fixture of the post-processing path (tests/golden/make_golden_post.py): oracle and the product's host entry
    points reproduce it bit for bit."""
    import zlib
    from dr_slam_amd import lib
    g = np.load(GOLD_POST)
    dm = oracle_mod.depth_to_float(g["depth"], np.float32(1.0) / g["depth_factor"])
    cloud, nrm = oracle_mod.post_surface_normals(dm, g["K4"], float(g["max_point_dist"]))
    assert zlib.crc32(cloud.tobytes()) == int(g["cloud_crc"])
    nan = np.isnan(g["normals"])
    assert np.array_equal(np.isnan(nrm), nan) and np.array_equal(nrm.view(np.uint32)[~nan], g["normals"].view(np.uint32)[~nan])
    for vg in (oracle_mod.post_voxel_grid, lib.plane_voxel_grid):
        assert np.array_equal(vg(g["plane_points"], 0.05).view(np.uint32), g["voxels"].view(np.uint32))
    for rf in (oracle_mod.post_refit, lib.plane_refit):
        ok, coef = rf(g["coef0"], g["voxels"], float(g["refit_threshold"]))
        assert ok == bool(g["refit_valid"]) and np.array_equal(coef.view(np.uint32), g["refit_coef"].view(np.uint32))
