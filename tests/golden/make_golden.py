#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.

The reference holds no golden vector, KAT or fixture for this path and cannot be
built or imported here, so these fixtures pin the *oracle's* outputs on committed inputs: any change to
oracle or kernels that moves a bit fails the golden tests.  Each file holds the input frame(s) and the
expected outputs (data only).
"""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dr_slam_amd import synth
from oracle import oracle as orc

OUT = os.path.dirname(os.path.abspath(__file__))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def orb_case(name, gray, params):
    o = orc.OrbOracle(*params)
    kps, desc = o(gray)
    nl = params[2]
    np.savez_compressed(os.path.join(OUT, name), gray=gray, params=np.array(params, np.float64), kps=kps, desc=desc,
                        pyr_crc=np.array([crc(o.pyramid(l)) for l in range(nl)], np.uint32),
                        blur_crc=np.array([crc(o.blurred(l)) if o.blurred(l) is not None else 0 for l in range(nl)], np.uint32),
                        cand_crc=np.array([crc(o.candidates(l)) for l in range(nl)], np.uint32),
                        cand_n=np.array([len(o.candidates(l)) for l in range(nl)], np.int32))
    return o, kps, desc


# config 1: low-texture 640x480 frame (exercises the minThFAST fallback), default ORB parameters
g, _, _ = next(synth.sequence(1, 1, cam=synth.TUM3, kind="planar_lowtexture"))
orb_case("orb_lowtexture_640x480.npz", g, (1000, 1.2, 8, 20, 7))

# small textured frame, non-default parameters
cam = synth.TUM3.scaled(0.5)
g, _, _ = next(synth.sequence(4, 1, cam=cam))
orb_case("orb_room_320x240.npz", g, (500, 1.2, 6, 20, 7))

# config 2: two consecutive 320x240 RGB-D frames, extract + stereo/grid + SearchByProjection(th=15)
fr = list(synth.sequence(2, 2, cam=cam))
o = orc.OrbOracle(500, 1.2, 6, 20, 7)
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
fo = []
for gray, depth, _ in fr:
    kps, desc = o(gray)
    fo.append(orc.FrameOracle(kps, desc, orc.depth_to_float(depth, np.float32(1) / np.float32(cam.depth_factor)), K4,
                              cam.bf, cam.w, cam.h, o.scale))
Twc = np.stack([f[2] for f in fr])
Tcw = np.linalg.inv(Twc).astype(np.float32)
Twc = Twc.astype(np.float32)
world, valid = fo[0].unproject(Twc[0])
mp = np.zeros(fo[0].N, orc.MAPPOINT_DTYPE)
mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, fo[0].desc
n, m = orc.search_by_projection_last(fo[1], fo[0], Tcw[1], Tcw[0], mp, 15.0, False, True)
off, idx = fo[1].grid_csr()
np.savez_compressed(os.path.join(OUT, "match_room_320x240.npz"), gray=np.stack([f[0] for f in fr]),
                    depth=np.stack([f[1] for f in fr]), Tcw=Tcw, Twc=Twc,
                    cam=np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.depth_factor, cam.w, cam.h], np.float64),
                    params=np.array((500, 1.2, 6, 20, 7), np.float64), kps0=fo[0].kps, kps1=fo[1].kps,
                    uRight1=fo[1].uRight, depth1=fo[1].depth, grid_off1=off, grid_idx1=idx, matches=m,
                    nmatches=np.int32(n))
for f in sorted(os.listdir(OUT)):
    if f.endswith(".npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))
