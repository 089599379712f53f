#!/usr/bin/env python3
"""Regenerate tests/golden/planes_lines_bow.npz: inputs (one 640x480 synthetic gray + depth frame,
descriptors of 300 ORB keypoints) and the oracle's outputs for the plane (AHC, CAPE), line (LSD+LBD)
and bag-of-words paths.  The reference has no fixtures for these paths; this pins the
oracle's own behaviour.  Data only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dr_slam_amd import synth, vocabulary as V
from oracle import oracle as orc

cam = synth.TUM3
g, d, _ = next(synth.sequence(3, 1, cam=cam, kind="living_room"))
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
f = np.float32(1.0) / np.float32(cam.depth_factor)
a = orc.ahc_planes(d, K4, f)
c = orc.cape_planes(orc.depth_to_float(d, f), K4, 20)
ln = orc.extract_lines(g)                     # lsd.cpp as OpenCV 3.4 spells it (rect_nfa's integer corners / quotients, nfa()'s `double(n) + 1`): the default
ln1 = orc.extract_lines(g, rect_mode=1)       # the LSD paper's reading of both (rounds 2-3)
ln2 = orc.extract_lines(g, rect_mode=2)       # integer corners with log_gamma(n + 1) (round 4's default)
kps, desc = orc.OrbOracle()(g)
desc = desc[:300]
ov = orc.VocabularyOracle(V.make_synthetic(6, 3, seed=2).to_text())
w, wt, nid = ov.transform_each(desc, 2)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "planes_lines_bow.npz")
np.savez_compressed(out, gray=g, depth=d, K4=K4, factor=f, ahc_planes=a["planes"], ahc_N=a["N"], ahc_seg=a["seg"],
                    cape_planes=c["planes"], cape_seg=c["seg"], lines=ln["lines"], ldesc=ln["desc"], lines_real=ln1["lines"], ldesc_real=ln1["desc"], lines_r4=ln2["lines"], ldesc_r4=ln2["desc"], orb_desc=desc,
                    bow_word=w, bow_nid=nid)
print(out, os.path.getsize(out), "planes", len(a["planes"]), len(c["planes"]), "lines", len(ln["lines"]))
