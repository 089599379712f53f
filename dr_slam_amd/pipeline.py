"""Host-side mirror of the reference's per-frame feature path, on top of the C-ABI (dr_slam_amd.lib).

Names and argument meaning follow the reference classes so parity tests read like reference code:

  ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)   include/ORBextractor.h:56-57
      __call__(image, mask) -> keypoints, descriptors                   src/ORBextractor.cc:1043
      GetLevels / GetScaleFactor / GetScaleFactors / ...                 include/ORBextractor.h:63-83
      mvImagePyramid                                                     include/ORBextractor.h:85
  ORBmatcher(nnratio, checkOri)                                          include/ORBmatcher.h:41
      SearchByProjection(CurrentFrame, LastFrame, th, bMono)             src/ORBmatcher.cc:1396
      SearchByProjection(F, vpMapPoints, th)                             src/ORBmatcher.cc:46
  FrontEnd: the batched, device-resident form used for throughput (one call = a batch of frames of one
      sequence: extract -> stereo/grid -> match each frame against its predecessor).

PyTorch appears only as the owner of device buffers / streams handed to the C-ABI as raw pointers.
"""
from __future__ import annotations

import numpy as np

from . import lib


class ORBextractor:
    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, max_width=640,
                 max_height=480, max_batch=1, device=0):
        self.ctx = lib.Context(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height,
                               max_batch, device)
        self.nlevels = nlevels
        self.scaleFactor = float(np.float32(scaleFactor))
        (self.mvScaleFactor, self.mvInvScaleFactor, self.mvLevelSigma2, self.mvInvLevelSigma2) = self.ctx.scale_tables()

    def __call__(self, image, mask=None):
        """Mask is ignored, as in the reference (include/ORBextractor.h:63 comment)."""
        return self.ctx.orb_extract(None if image is None else np.ascontiguousarray(image))

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self.mvScaleFactor

    def GetInverseScaleFactors(self):
        return self.mvInvScaleFactor

    def GetScaleSigmaSquares(self):
        return self.mvLevelSigma2

    def GetInverseScaleSigmaSquares(self):
        return self.mvInvLevelSigma2

    @property
    def mvImagePyramid(self):
        """Bordered pyramid of the most recent frame (slot 0); level l interior = [19:-19, 19:-19]."""
        return [self.ctx.pyramid_level(0, l) for l in range(self.nlevels)]


class ORBmatcher:
    TH_HIGH = 100
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    def SearchByProjectionLast(self, fe: "FrontEnd", cur_slot, last_slot, Tcw_cur, Tcw_last, last_mp, n_cur, th,
                               bMono=False, cur_mp=None, cur_obs=None):
        return fe.ctx.search_by_projection_last(cur_slot, last_slot, Tcw_cur, Tcw_last, fe.cam, last_mp, n_cur, th,
                                                bMono, self.mbCheckOrientation, cur_mp, cur_obs)

    def SearchByProjectionMap(self, fe: "FrontEnd", slot, vpMapPoints, n, th, frame_mp=None, claim_obs=None):
        return fe.ctx.search_by_projection_map(slot, vpMapPoints, n, th, self.mfNNratio, frame_mp, claim_obs)

    @staticmethod
    def DescriptorDistance(a, b):
        return int(np.unpackbits(np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))).sum())


class FrontEnd:
    """Batched extract + glue + match over device-resident frames of one sequence."""

    def __init__(self, cam, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, max_batch=8,
                 device=0, ctx=None):
        """ctx: an existing lib.Context to work on (one of a lib.Pipeline's) instead of a new one"""
        self.ctx = ctx if ctx is not None else lib.Context(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, cam.w, cam.h,
                                                           max_batch, device)
        self.cam = lib.make_camera(cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, cam.depth_factor, cam.w, cam.h)
        dist = tuple(getattr(cam, "dist", ()) or ())
        if dist and dist[0] != 0.0:
            # Frame::ComputeImageBounds + UndistortKeyPoints (src/Frame.cc:835-891): bounds of the undistorted
            # corners, mvKeysUn built on the device by every glue call
            b = self.ctx.image_bounds(self.cam, dist, cam.w, cam.h)
            self.cam.min_x, self.cam.max_x, self.cam.min_y, self.cam.max_y = (float(v) for v in b)
            self.ctx.set_distortion(self.cam, dist)
        self.w, self.h = cam.w, cam.h
        self.max_batch = max_batch

    def process(self, gray_t, depth_t, Tcw=None, Twc=None, th=15.0, check_ori=True, stream: int = 0):
        """gray_t: uint8 [B,H,W] CUDA tensor; depth_t: uint16 (or int16 view) [B,H,W] CUDA tensor or None.
        Asynchronous on `stream` (raw hipStream_t handle, 0 = context stream)."""
        B = int(gray_t.shape[0])
        assert gray_t.is_cuda and gray_t.is_contiguous() and gray_t.element_size() == 1
        self.ctx.orb_extract_batch_ptr(gray_t.data_ptr(), self.w * self.h, self.w, self.w, self.h, B, stream)
        if depth_t is not None:
            assert depth_t.is_cuda and depth_t.is_contiguous() and depth_t.element_size() == 2
            self.ctx.stereo_grid_batch_ptr(depth_t.data_ptr(), self.w * self.h, self.w, self.cam, B, stream)
            if Tcw is not None and B >= 2:
                self.ctx.match_consecutive_batch(Tcw, Twc, self.cam, th, False, check_ori, B, stream)
        return B

    def keypoints(self, slot):
        return self.ctx.orb_download(slot)

    def matches(self, slot):
        return self.ctx.match_download(slot)
