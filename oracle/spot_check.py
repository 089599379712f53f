"""oracle/spot_check.py - TEST INFRASTRUCTURE (see oracle.h): the checker half of the bench-size parity check.

A batch that went through the headline path (FrontEnd.process / drfe_pipeline_submit: ORB extract -> Undistort / Stereo / Grid
glue -> SearchByProjection of every frame against its predecessor) is compared, slot by slot, with the CPU oracle on the
same input frames: keypoint records and 256-bit descriptors of slot s and of slot s - 1, and the match array of slot s
(reference src/ORBextractor.cc:1043-1105, src/Frame.cc:224-237,835-911, src/ORBmatcher.cc:1396-1535).  Bar: identical bytes.

Used by tests/ (pytest -m gpu at the bench's batch size) and by bench.py AFTER its timed loop; nothing under dr_slam_amd/
imports this module.  The oracle is the unpinned restatement DESIGN.md section 5 describes."""
import numpy as np

from . import oracle as orc


class SlotChecker:
    """oracle frames are cached per distinct input frame (bench batches repeat a short sequence ping-pong)"""

    def __init__(self, cam):
        self.cam = cam
        self.orb = orc.OrbOracle()
        self.K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        self.inv = np.float32(1.0) / np.float32(cam.depth_factor)
        self.cache = {}

    def frame(self, key, gray, depth16):
        fo = self.cache.get(key)
        if fo is None:
            kps, desc = self.orb(gray)
            fo = orc.FrameOracle(kps, desc, orc.depth_to_float(depth16, self.inv), self.K4, self.cam.bf, self.cam.w, self.cam.h,
                                 self.orb.scale, dist=getattr(self.cam, "dist", None))
            fo.kps_raw = kps
            self.cache[key] = fo
        return fo

    def check(self, fe, gray, depth16, Tcw, Twc, slots, keys=None, th=15.0, check_ori=True, what=""):
        """fe: a dr_slam_amd.pipeline.FrontEnd whose context holds the processed batch; gray [B,H,W] u8, depth16 [B,H,W] u16
        (host arrays of what was processed); slots: slot numbers >= 1; keys[s]: identity of slot s's input frame (for the
        cache; default: the slot number).  Raises AssertionError on the first mismatch; returns the number of slots checked."""
        n = 0
        for s in slots:
            s = int(s)
            assert s >= 1
            pair = []
            for t in (s - 1, s):
                fo = self.frame(t if keys is None else keys[t], gray[t], depth16[t])
                kps, desc = fe.keypoints(t)
                assert len(kps) == fo.N, f"{what}slot {t}: {len(kps)} keypoints, oracle {fo.N}"
                assert np.array_equal(kps.view(np.uint8), fo.kps_raw.view(np.uint8)), f"{what}slot {t}: keypoint records differ from the oracle"
                assert np.array_equal(desc, fo.desc), f"{what}slot {t}: descriptors differ from the oracle"
                pair.append(fo)
            last, cur = pair
            world, valid = last.unproject(Twc[s - 1])
            mp = np.zeros(last.N, orc.MAPPOINT_DTYPE)
            mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, last.desc
            n_o, m_o = orc.search_by_projection_last(cur, last, Tcw[s], Tcw[s - 1], mp, th, False, check_ori)
            m_g, n_g = fe.matches(s)
            assert n_g == n_o and np.array_equal(m_g[:cur.N], m_o), f"{what}slot {s}: match array differs from the oracle ({n_g} vs {n_o} matches)"
            n += 1
        return n
