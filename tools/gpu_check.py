#!/usr/bin/env python3
"""First-light diagnostic on the GPU box: stage-by-stage comparison of the HIP path with the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dr_slam_amd import lib, synth
from oracle import oracle as orc

frames = list(synth.sequence(2, 3))
ctx = lib.Context(max_batch=4)
o = orc.OrbOracle()
for fi, (g, d, T) in enumerate(frames[:2]):
    t = time.time(); kps, desc = ctx.orb_extract(g); t1 = time.time() - t
    okps, odesc = o(g)
    print(f"frame {fi}: gpu N={len(kps)} oracle N={len(okps)}  ({t1*1e3:.1f} ms)")
    for l in range(8):
        p = ctx.pyramid_level(0, l); op = o.pyramid(l)
        b = ctx.blurred_level(0, l); ob = o.blurred(l)
        c = ctx.candidates(0, l); oc = o.candidates(l)
        same_c = c.shape == oc.shape and np.array_equal(c, oc)
        print(f"  L{l}: pyr {'OK' if np.array_equal(p, op) else 'DIFF %d' % (p != op).sum()}"
              f"  blur {'OK' if ob is None or np.array_equal(b, ob) else 'DIFF %d' % (b != ob).sum()}"
              f"  cand gpu {len(c)} oracle {len(oc)} {'OK' if same_c else 'DIFF'}")
        if not same_c and len(c) and len(oc):
            sc = set(map(tuple, c.tolist())); so = set(map(tuple, oc.tolist()))
            print("     only gpu:", sorted(sc - so)[:5], " only oracle:", sorted(so - sc)[:5])
    n = min(len(kps), len(okps))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        bad = np.nonzero(kps[f][:n] != okps[f][:n])[0]
        print(f"  kp.{f}: {'OK' if len(bad) == 0 and len(kps) == len(okps) else 'DIFF at %s' % bad[:8]}")
        if len(bad): print("     gpu", kps[f][bad[:4]], "oracle", okps[f][bad[:4]])
    bad = np.nonzero((desc[:n] != odesc[:n]).any(1))[0]
    print(f"  desc: {'OK' if len(bad) == 0 else 'DIFF rows %s' % bad[:8]}")
