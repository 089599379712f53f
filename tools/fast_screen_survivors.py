"""How many pixels would a FAST early reject let through?  For the three textures of the bench / tests, over all pyramid levels:
 - score >= minTh (7): the pixels whose exact corner score can matter (everything else may be scored 0);
 - the compass screen (necessary condition from the ring positions 0, 4, 8, 12: two adjacent ones brighter / darker by minTh);
 - the quad screen (every 9-arc holds two of the eight aligned 4-windows: the bound max_m min4(m) the full tree already has);
at pixel granularity and at the granularity of the kernel's pixel pairs (a pair survives if either pixel does).
CPU only (numpy): python tools/fast_screen_survivors.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import synth
from oracle import oracle as O

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def ring_stack(img):
    h, w = img.shape
    c = img[3:h - 3, 3:w - 3].astype(np.int16)
    d = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int16) for dx, dy in RING])
    return c, d


def score(c, d):
    dd = d - c
    best = np.zeros(c.shape, np.int16)
    ext = np.concatenate([dd, dd[:8]])
    for s in range(16):
        arc = ext[s:s + 9]
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    return best


def main(TH=7):
    o = O.OrbOracle()
    for kind, cam in (("room_boxes", synth.TUM3), ("living_room", synth.ICL), ("planar_lowtexture", synth.TUM3)):
        g = next(synth.sequence(2, 1, cam=cam, kind=kind))[0]
        o(g)                                              # extract once: the oracle keeps the pyramid
        tot = np.zeros(7)
        levels = [o.pyramid(l)[16:-16, 16:-16] for l in range(o.nlevels)]      # detection area + the 3-pixel ring margin
        for img in levels:
            img = np.asarray(img)
            c, d = ring_stack(img)
            s = score(c, d)
            dd = d - c
            comp = dd[[0, 4, 8, 12]]
            nxt = np.roll(comp, -1, axis=0)
            ub = np.maximum(np.minimum(comp, nxt).max(0), np.minimum(-comp, -nxt).max(0))
            q4 = np.stack([np.minimum.reduce([dd[(2 * m + 1 + k) % 16] for k in range(4)]) for m in range(8)])
            q4d = np.stack([np.minimum.reduce([-dd[(2 * m + 1 + k) % 16] for k in range(4)]) for m in range(8)])
            uq = np.maximum(q4.max(0), q4d.max(0))
            W = (c.shape[1] // 2) * 2

            def pair(m):
                m = m[:, :W]
                return (m[:, 0::2] | m[:, 1::2]).mean()
            a, b, q = s >= TH, ub >= TH, uq >= TH
            n = c.size
            tot += np.array([n, a.sum(), b.sum(), q.sum(), pair(a) * n, pair(b) * n, pair(q) * n])
        n = tot[0]
        print("%-18s pixels %8d: score>=TH %5.1f%%  compass screen %5.1f%%  quad screen %5.1f%% | pairs: score %5.1f%%  compass %5.1f%%  quad %5.1f%%"
              % (kind, n, 100 * tot[1] / n, 100 * tot[2] / n, 100 * tot[3] / n, 100 * tot[4] / n, 100 * tot[5] / n, 100 * tot[6] / n))


if __name__ == "__main__":
    for th in (7, 20):            # minThFAST and iniThFAST (ORBextractor.cc:809-815)
        print("threshold", th)
        main(th)
