#!/usr/bin/env python3
"""Regenerate tests/golden/post_room_320x240.npz: the plane post-processing / surface-normal path (rows a-20, f-2) on one
committed 320x240 depth frame.  Like the other fixtures it pins the ORACLE's outputs (the reference holds no vector for
this path and PCL cannot be built here): a change of oracle or kernels that moves a bit fails the golden tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dr_slam_amd import synth          # noqa: E402
from oracle import oracle as orc       # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
cam = synth.TUM3.scaled(0.5)
_, d, _ = next(synth.sequence(2, 1, cam=cam))
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
dm = orc.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor))
cloud, nrm = orc.post_surface_normals(dm, K4, 9.0)
# a plane cloud for the voxel grid / refit: the room's floor (y = 1.4 m below the camera) in the 3x-subsampled cloud
pts = cloud.reshape(-1, 3)
pts = pts[(pts[:, 2] > 0) & (np.abs(pts[:, 1] - 1.4) < 0.03)]
vox = orc.post_voxel_grid(pts, 0.05)
c = pts.astype(np.float64).mean(0)
u, s, vt = np.linalg.svd(pts.astype(np.float64) - c)
n0 = vt[2] if vt[2] @ c <= 0 else -vt[2]
coef0 = np.array([*n0, -(n0 @ c)], np.float32)
ok, coef = orc.post_refit(coef0, vox, 0.05)
np.savez_compressed(os.path.join(OUT, "post_room_320x240.npz"), depth=d, K4=K4, depth_factor=np.float32(cam.depth_factor),
                    max_point_dist=np.float32(9.0), normals=nrm, cloud_crc=np.uint32(__import__("zlib").crc32(cloud.tobytes())),
                    plane_points=pts, voxels=vox, coef0=coef0, refit_threshold=np.float64(0.05), refit_valid=np.bool_(ok),
                    refit_coef=coef)
print("post_room_320x240.npz", os.path.getsize(os.path.join(OUT, "post_room_320x240.npz")), "voxels", len(vox), "valid", ok)
