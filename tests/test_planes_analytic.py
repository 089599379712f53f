"""Independent checks of the host-restated plane stages (CAPE, PEAC / AHC, PCL-style surface normals) on analytic scenes with
closed-form answers (tests/plane_scenarios.py): the extractors must find the three planes of a rendered room corner where they
were put - normals within a stated angle, distances within a stated error, the label partition equal to the ground truth away
from a stated boundary band - and the normals of a tilted plane must be its normal.  The CPU tests hold the ORACLE to the ground
truth, the -m gpu tests the PRODUCT (which the parity tests elsewhere hold to the oracle bit for bit): a shared misreading of
CAPE / PEAC / PCL that moved a plane or a boundary would fail here."""
import numpy as np
import pytest

from plane_scenarios import room_corner, tilted_plane, check_planes


def _cam():
    from dr_slam_amd import synth
    return synth.TUM3


def _k4(cam):
    return np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)


def _inv(cam):
    return float(np.float32(1.0) / np.float32(cam.depth_factor))


# stated tolerances: AHC fits 0.5 deg / 1.5 cm / labels exact beyond 1 px of a true boundary (it refines per pixel);
# CAPE 1.5 deg / 5 cm (float32 cell sums, the grazing floor) / labels exact beyond 4 px (cells of 20 px, eroded then refined)
AHC_TOL = dict(max_angle_deg=0.5, max_d_err=0.015, band_px=1)
CAPE_TOL = dict(max_angle_deg=1.5, max_d_err=0.05, band_px=4)


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_ahc_finds_the_analytic_planes(oracle_mod, seed):
    cam = _cam()
    planes, lab, _, d16 = room_corner(cam, noise=2, seed=seed)
    r = oracle_mod.ahc_planes(d16, _k4(cam), _inv(cam))
    normals, centers = r["planes"][:, :3], r["planes"][:, 3:6]
    check_planes(normals, np.einsum("ij,ij->i", normals, centers), r["seg"], planes, lab, **AHC_TOL)


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_cape_finds_the_analytic_planes(oracle_mod, seed):
    cam = _cam()
    planes, lab, _, d16 = room_corner(cam, noise=2, seed=seed)
    c = oracle_mod.cape_planes(d16.astype(np.float32) * np.float32(_inv(cam)), _k4(cam))
    check_planes(c["planes"][:, :3], c["planes"][:, 6], c["seg"], planes, lab, **CAPE_TOL)


def test_oracle_surface_normals_of_a_tilted_plane(oracle_mod):
    """pcl::IntegralImageNormalEstimation (AVERAGE_3D_GRADIENT) on an exact plane: every finite normal is the plane's normal to
    3e-4 rad (float32 depth and cloud), the median to 1e-5, and normals exist away from the image border."""
    cam = _cam()
    z, n = tilted_plane(cam)
    cloud, nrm = oracle_mod.post_surface_normals(z, _k4(cam), 9.0)
    fin = np.isfinite(nrm).all(-1)
    assert fin.mean() > 0.7
    dots = np.abs(nrm[fin].astype(np.float64) @ n)
    assert np.arccos(np.clip(dots.min(), 0, 1)) < 3e-4 and np.arccos(np.clip(np.median(dots), 0, 1)) < 1e-5
    # the reference flips every normal towards the camera (the origin): n . p < 0 for the points of the cloud
    assert (np.einsum("ij,ij->i", nrm[fin].astype(np.float64), cloud[fin].astype(np.float64)) < 0).all()


@pytest.mark.gpu
def test_product_plane_extractors_find_the_analytic_planes():
    from dr_slam_amd import lib
    cam = _cam()
    planes, lab, _, d16 = room_corner(cam, noise=2, seed=0)
    ctx = lib.Context(max_batch=1)
    try:
        a = ctx.planes_ahc(d16, _k4(cam), _inv(cam))
        normals = np.asarray(a["planes"]["normal"], np.float64)
        centers = np.asarray(a["planes"]["center"], np.float64)
        check_planes(normals, np.einsum("ij,ij->i", normals, centers), a["seg"], planes, lab, **AHC_TOL)
        c = ctx.planes_cape(d16.astype(np.float32) * np.float32(_inv(cam)), _k4(cam))
        check_planes(np.asarray(c["planes"]["normal"], np.float64), np.asarray(c["planes"]["d"], np.float64), c["seg"], planes, lab, **CAPE_TOL)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_product_surface_normals_of_a_tilted_plane():
    from dr_slam_amd import lib
    cam = _cam()
    z, n = tilted_plane(cam)
    ctx = lib.Context(max_batch=1)
    try:
        recs = ctx.surface_normals(z, _k4(cam), 9.0)
        nr = recs["normal"].astype(np.float64)
        fin = np.isfinite(nr).all(-1)
        assert len(recs) > 5000 and fin.mean() > 0.6
        dots = np.abs(nr[fin] @ n)
        assert np.arccos(np.clip(dots.min(), 0, 1)) < 3e-4 and np.arccos(np.clip(np.median(dots), 0, 1)) < 1e-5
    finally:
        ctx.close()
