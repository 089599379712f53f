"""Regenerates tests/golden/lsd_projection.npz: expected outputs of the CPU oracle for the seeded
LSDmatcher::SearchByProjection scenarios of tests/line_scenarios.py.  The reference's own
implementation cannot run here (OpenCV/Eigen absent), so this pins the ORACLE against drift, not the reference.
Run from the repo root:  python tests/golden/make_golden3.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import line_scenarios as LS  # noqa: E402
from oracle import oracle as O  # noqa: E402

KL = np.dtype([("pt_x", "<f4"), ("pt_y", "<f4"), ("angle", "<f4"), ("octave", "<i4")])
CASES = [(1, 0.0, 15.0), (2, 0.5, 15.0), (3, -0.5, 15.0), (4, 0.0, 7.0), (5, 0.0, 30.0)]

out = {}
for seed, motion, th in CASES:
    sc = LS.make(seed, KL, O.MAPLINE_DTYPE, O.TRACKED_LINE_DTYPE, motion=motion)
    n, ml = O.lsd_search_by_projection_last(LS.cam9(), sc["Tcw_cur"], sc["Tcw_last"], LS.SCALE, sc["last"], sc["cur"],
                                            sc["cur_desc"], th, False, 0.9, sc["cur_ml"], sc["cur_obs"])
    n2, ml2 = O.lsd_search_by_projection_map(LS.SCALE, sc["tracked"], sc["cur"], sc["cur_desc"], th / 15.0, 0.9, sc["cur_ml"],
                                             sc["cur_obs"])
    out[f"last_{seed}"] = np.concatenate([[n], ml]).astype(np.int32)
    out[f"map_{seed}"] = np.concatenate([[n2], ml2]).astype(np.int32)
out["cases"] = np.array(CASES, np.float64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lsd_projection.npz"), **out)
print({k: int(v[0]) for k, v in out.items() if k != "cases"})
