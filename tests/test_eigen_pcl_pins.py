"""Oracle vs the REAL Eigen 3.3.7 / PCL 1.9, when somebody has produced tests/golden/eigen_pcl_pins.bin with
tools/dump_eigen_pcl_reference.cpp on a machine that has those libraries (the build container has neither, and no network): every
test here then SKIPS with the reason "parity unpinned" - the honest state of the Eigen / PCL-level parity (SURVEY.md section 8c,
DESIGN.md section 5).  The dumper's inputs come from SplitMix64, regenerated here in integer arithmetic."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# ---- Eigen 3.3.7 / PCL 1.9 (tools/dump_eigen_pcl_reference.cpp, compiled and run where those libraries exist) -------------------

EPINS = os.path.join(GOLD, "eigen_pcl_pins.bin")
needs_epins = pytest.mark.skipif(not os.path.exists(EPINS), reason="parity unpinned: tests/golden/eigen_pcl_pins.bin absent (build and run "
                                 "tools/dump_eigen_pcl_reference.cpp where Eigen 3.3.7 and PCL 1.9 are installed)")


class _SplitMix:
    """the generator of tools/dump_eigen_pcl_reference.cpp, integer arithmetic only"""
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def unit(self):
        m = 0xFFFFFFFFFFFFFFFF
        self.s = (self.s + 0x9E3779B97F4A7C15) & m
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        z ^= z >> 31
        return float(z >> 11) * (1.0 / 9007199254740992.0)


def _sections():
    raw = open(EPINS, "rb").read()
    out, o = {}, 0
    assert raw[o:o + 4] == b"EIG3"; o += 4
    n = int(np.frombuffer(raw, np.int32, 1, o)[0]); o += 4
    out["eig"] = np.frombuffer(raw, np.float64, 21 * n, o).reshape(n, 21); o += 8 * 21 * n
    assert raw[o:o + 4] == b"VOXG"; o += 4
    n = int(np.frombuffer(raw, np.int32, 1, o)[0]); o += 4
    out["vox_in"] = np.frombuffer(raw, np.float32, 3 * n, o).reshape(n, 3); o += 12 * n
    m = int(np.frombuffer(raw, np.int32, 1, o)[0]); o += 4
    out["vox_out"] = np.frombuffer(raw, np.float32, 3 * m, o).reshape(m, 3); o += 12 * m
    assert raw[o:o + 4] == b"SACP"; o += 4
    out["sac_start"] = np.frombuffer(raw, np.float32, 4, o); o += 16
    out["sac_valid"] = int(np.frombuffer(raw, np.int32, 1, o)[0]); o += 4
    out["sac_coef"] = np.frombuffer(raw, np.float32, 4, o); o += 16
    assert raw[o:o + 4] == b"NORM"; o += 4
    W, H = (int(v) for v in np.frombuffer(raw, np.int32, 2, o)); o += 8
    out["org"] = np.frombuffer(raw, np.float32, 3 * W * H, o).reshape(H, W, 3); o += 12 * W * H
    out["nrm"] = np.frombuffer(raw, np.float32, 3 * W * H, o).reshape(H, W, 3); o += 12 * W * H
    assert o == len(raw)
    return out


@needs_epins
def test_eigen_selfadjoint_3x3_equal_eigen(oracle_mod):
    """LA::eig33sym (reference include/peac/eig33sym.hpp:70-74) = Eigen::SelfAdjointEigenSolver<Matrix3d>::compute: eigenvalues and
    eigenvectors of 4096 symmetric matrices (scaled, degenerate ones among them), bit for bit."""
    for rec in _sections()["eig"]:
        s, V = oracle_mod.eig33sym(rec[:9].reshape(3, 3))
        assert np.array_equal(s.view(np.uint64), rec[9:12].view(np.uint64))
        assert np.array_equal(V.reshape(-1).view(np.uint64), rec[12:21].view(np.uint64))


@needs_epins
def test_pcl_voxel_grid_and_refit_equal_pcl(oracle_mod):
    """pcl::VoxelGrid(0.05) (leaf order = std::sort's permutation, float centroid sums in that order) and the refit of
    Frame::MaxPointDistanceFromPlane (SACSegmentation RANSAC + optimizeModelCoefficients) on the dumper's cloud."""
    S = _sections()
    got = oracle_mod.post_voxel_grid(S["vox_in"], 0.05)
    assert got.shape == S["vox_out"].shape and np.array_equal(got.view(np.uint32), S["vox_out"].view(np.uint32))
    valid, coef = oracle_mod.post_refit(S["sac_start"], S["vox_out"], 0.10)
    assert int(valid) == S["sac_valid"]
    if valid:
        # MaxPointDistanceFromPlane flips the refitted plane to the sign of the start plane's d; the dump holds PCL's own sign
        ref = S["sac_coef"] if (S["sac_coef"][3] < 0) == (S["sac_start"][3] < 0) else -S["sac_coef"]
        assert np.array_equal(np.asarray(coef, np.float32).view(np.uint32), ref.view(np.uint32))


@needs_epins
def test_pcl_integral_image_normals_equal_pcl(oracle_mod):
    """pcl::IntegralImageNormalEstimation (AVERAGE_3D_GRADIENT, 0.05, 10) on the dumper's organized cloud: rebuilt here as the
    depth image whose 3x-subsampled cloud it is."""
    S = _sections()
    H, W, _ = S["org"].shape
    g = _SplitMix(0x1122334455667788)
    depth = np.zeros((3 * H, 3 * W), np.float32)
    for r in range(H):
        for c in range(W):
            f = np.float32
            z = f(f(1.5) + f(f(0.004) * f(c))) + f(f(0.002) * f(r)) if c < 60 else f(1.74) + f(f(0.02) * f(c - 60))
            z = f(z) + f(0.001 * (2.0 * g.unit() - 1.0))
            if 30 < r < 36 and 20 < c < 30:
                z = f(0)
            depth[3 * r, 3 * c] = z
    cloud, nrm = oracle_mod.post_surface_normals(depth[:3 * H - 2, :3 * W - 2], np.array([260, 260, 160, 120], np.float32), 1e9)
    assert np.array_equal(cloud.view(np.uint32), S["org"].view(np.uint32))
    assert np.array_equal(np.isnan(nrm), np.isnan(S["nrm"]))
    ok = ~np.isnan(nrm)
    assert np.array_equal(nrm[ok].view(np.uint32), S["nrm"][ok].view(np.uint32))
