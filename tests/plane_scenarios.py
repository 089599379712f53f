"""Analytic depth scenes with closed-form answers for the plane extractors and the surface normals (the counterpart of
line_scenarios.py): a room corner of three planes at known poses, rendered by ray / plane intersection, and a single tilted
plane.  Nothing here comes from the oracle or the product: it is the independent check VERDICT round 2 asked for (the oracle and
the host-restated plane stages share one author's reading of CAPE / PEAC / PCL)."""
import numpy as np


def _rays(cam):
    u, v = np.meshgrid(np.arange(cam.w), np.arange(cam.h))
    return np.stack([(u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u, float)], -1)


def room_corner(cam, noise=2, seed=0):
    """Floor y = 1 m, back wall (normal 10 deg off the optical axis, 3 m), left wall (8 deg, 1.3 m).  Returns
    (planes [(unit normal, d with n . p = d)], label image of the visible plane, depth in metres, uint16 depth with +-noise
    counts of sensor noise at the camera's depth factor)."""
    a, b = np.deg2rad(10.0), np.deg2rad(8.0)
    planes = [(np.array([0.0, 1.0, 0.0]), 1.0),
              (np.array([np.sin(a), 0.0, np.cos(a)]), 3.0),
              (np.array([-np.cos(b), 0.0, np.sin(b)]), 1.3)]
    dirs = _rays(cam)
    zs = []
    for n, d in planes:
        den = dirs @ n
        zs.append(np.where(den > 1e-9, d / np.where(den > 1e-9, den, 1.0), np.inf))
    zs = np.stack(zs)
    lab, z = zs.argmin(0), zs.min(0)
    rng = np.random.default_rng(seed)
    d16 = np.clip(np.rint(z * cam.depth_factor) + rng.integers(-noise, noise + 1, z.shape), 0, 65535).astype(np.uint16)
    return planes, lab, z, d16


def tilted_plane(cam, tilt_x_deg=20.0, tilt_y_deg=-12.0, dist=2.5):
    """One plane n . p = dist seen by every pixel: exact float32 depth (no sensor quantisation) and its unit normal."""
    n = np.array([np.sin(np.deg2rad(tilt_x_deg)), np.sin(np.deg2rad(tilt_y_deg)), 0.0])
    n[2] = np.sqrt(1.0 - n[0] ** 2 - n[1] ** 2)
    return (dist / (_rays(cam) @ n)).astype(np.float32), n


def boundary_band(lab, r):
    """Pixels within r of a change of the ground-truth label (r = 0: none)."""
    from scipy import ndimage
    b = np.zeros(lab.shape, bool)
    dx, dy = lab[:, 1:] != lab[:, :-1], lab[1:, :] != lab[:-1, :]
    b[:, 1:] |= dx; b[:, :-1] |= dx; b[1:, :] |= dy; b[:-1, :] |= dy
    return ndimage.binary_dilation(b, iterations=r) if r > 0 else np.zeros(lab.shape, bool)


def check_planes(normals, ds, seg, planes, lab, max_angle_deg, max_d_err, band_px):
    """normals [n, 3], ds [n] (|n . p|), seg (0 = none, i + 1 = plane i) against the ground truth: one extracted plane per true
    plane, normal within max_angle_deg (sign-free), distance within max_d_err metres, and every pixel farther than band_px from a
    true boundary labelled with its true plane."""
    assert len(normals) == len(planes), (len(normals), len(planes))
    match = [max(range(len(planes)), key=lambda k: abs(float(n @ planes[k][0]))) for n in normals]
    assert sorted(match) == list(range(len(planes))), match
    for n, d, k in zip(normals, ds, match):
        ang = np.degrees(np.arccos(min(1.0, abs(float(n @ planes[k][0])))))
        assert ang < max_angle_deg, (k, ang)
        assert abs(abs(d) - planes[k][1]) < max_d_err, (k, d, planes[k][1])
    pred = np.full(lab.shape, -1)
    for i, k in enumerate(match):
        pred[seg == i + 1] = k
    far = ~boundary_band(lab, band_px)
    assert not (far & (pred != lab)).any(), int((far & (pred != lab)).sum())
