"""ctypes wrapper over oracle/libdrfe_oracle.so — TEST INFRASTRUCTURE (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
MAPPOINT_DTYPE = np.dtype([("valid", "u1"), ("obsPositive", "u1"), ("pad", "u1", (2,)), ("world", "<f4", (3,)),
                           ("desc", "u1", (32,))])
TRACKED_DTYPE = np.dtype([("trackInView", "u1"), ("bad", "u1"), ("obsPositive", "u1"), ("pad", "u1"),
                          ("level", "<i4"), ("projX", "<f4"), ("projY", "<f4"), ("projXR", "<f4"),
                          ("viewCos", "<f4"), ("desc", "u1", (32,))])


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "libdrfe_oracle.so")
    if force or not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "-j4"], stdout=subprocess.DEVNULL)
    return path


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        # DRFE_ORACLE_LIB: bench.py's timing leg loads its own -march=native build (kept outside the tree)
        _LIB = C.CDLL(os.environ.get("DRFE_ORACLE_LIB") or build())
        L = _LIB
        L.orc_last_error.restype = C.c_char_p
        L.orc_orb_create.restype = C.c_void_p
        L.orc_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_orb_destroy.argtypes = [C.c_void_p]
        L.orc_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long]
        for n in ("orc_orb_get_keypoints", "orc_orb_get_descriptors"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_orb_geometry.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_orb_get_pyramid.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_orb_get_blurred.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_orb_num_candidates.argtypes = [C.c_void_p, C.c_int]
        L.orc_orb_get_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_long]
        L.orc_reflect101.argtypes = [C.c_int, C.c_int]
        L.orc_fast_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_score_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p]
        L.orc_gaussian_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_long]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.orc_orb_descriptor.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 5 + [C.c_void_p]
        L.orc_hamming_swar.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_depth_to_float.argtypes = [C.c_void_p, C.c_long, C.c_float, C.c_void_p]
        L.orc_frame_create.restype = C.c_void_p
        L.orc_frame_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_frame_destroy.argtypes = [C.c_void_p]
        L.orc_frame_get_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_frame_grid_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_frame_features_in_area.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_int]
        L.orc_frame_unproject.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_by_projection_last.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_search_by_projection_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                                   C.c_void_p, C.c_void_p]
        L.orc_bf_knn_hamming.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_match_orb_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ahc_run.restype = C.c_void_p
        L.orc_ahc_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float]
        L.orc_ahc_free.argtypes = [C.c_void_p]
        for n in ("orc_ahc_num_planes", "orc_ahc_num_blocks"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_ahc_get_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ahc_get_seg.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_ahc_member_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_ahc_get_members.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_ahc_get_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_eig33sym.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_cape_run.restype = C.c_void_p
        L.orc_cape_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.orc_cape_free.argtypes = [C.c_void_p]
        for n in ("orc_cape_num_planes", "orc_cape_num_cells"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_cape_get_planes.argtypes = [C.c_void_p] * 4
        L.orc_cape_get_seg.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_cape_get_cells.argtypes = [C.c_void_p] * 4
        L.orc_voc_create.restype = C.c_void_p
        L.orc_voc_create.argtypes = [C.c_char_p]
        L.orc_voc_free.argtypes = [C.c_void_p]
        L.orc_voc_info.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_voc_get_nodes.argtypes = [C.c_void_p] * 5
        L.orc_voc_transform_each.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_voc_transform_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_search_by_bow.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        assert L.orc_sizeof_keypoint() == KP_DTYPE.itemsize
        assert L.orc_sizeof_mappointrec() == MAPPOINT_DTYPE.itemsize
        assert L.orc_sizeof_trackedpointrec() == TRACKED_DTYPE.itemsize
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class OrbOracle:
    """Restatement of ORBextractor (reference include/ORBextractor.h:51-85)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_orb_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        sc, isc, s2, is2 = (np.zeros(nlevels, np.float32) for _ in range(4))
        q = np.zeros(nlevels, np.int32)
        um = np.zeros(16, np.int32)
        self.L.orc_orb_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(q), _p(um))
        self.scale, self.inv_scale, self.sigma2, self.inv_sigma2, self.quota, self.umax = sc, isc, s2, is2, q, um
        self.geom = None

    def __del__(self):
        try:
            self.L.orc_orb_destroy(self.h)
        except Exception:
            pass

    def geometry(self, w, h):
        g = np.zeros((self.nlevels, 11), np.int32)
        if self.L.orc_orb_geometry(self.h, w, h, _p(g)) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return g  # w,h,quota,minBX,minBY,maxBX,maxBY,nCols,nRows,wCell,hCell

    def __call__(self, gray):
        gray = _c(gray, np.uint8)
        hh, w = gray.shape
        n = self.L.orc_orb_extract(self.h, _p(gray), w, hh, gray.strides[0])
        if n < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        kps = np.zeros(n, KP_DTYPE)
        desc = np.zeros((n, 32), np.uint8)
        self.L.orc_orb_get_keypoints(self.h, _p(kps))
        self.L.orc_orb_get_descriptors(self.h, _p(desc))
        self.geom = self.geometry(w, hh)
        return kps, desc

    def pyramid(self, l):
        g = self.geom[l]
        out = np.zeros((g[1] + 38, g[0] + 38), np.uint8)
        self.L.orc_orb_get_pyramid(self.h, l, _p(out))
        return out

    def blurred(self, l):
        g = self.geom[l]
        out = np.zeros((g[1], g[0]), np.uint8)
        ok = self.L.orc_orb_get_blurred(self.h, l, _p(out))
        return out if ok else None

    def candidates(self, l):
        n = self.L.orc_orb_num_candidates(self.h, l)
        out = np.zeros((n, 3), np.int32)
        self.L.orc_orb_get_candidates(self.h, l, _p(out))
        return out

    def distribute(self, keys3, minX, maxX, minY, maxY, N):
        keys3 = _c(keys3, np.int32)
        out = np.zeros(max(len(keys3), 1), np.int32)
        n = self.L.orc_distribute_octtree(self.h, _p(keys3), len(keys3), minX, maxX, minY, maxY, N, _p(out))
        if n < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return out[:n]

    def ic_angle(self, img, x, y):
        img = _c(img, np.uint8)
        return self.L.orc_ic_angle(self.h, _p(img), img.strides[0], x, y)


def resize_linear(src, dw, dh):
    src = _c(src, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(out), dw, dh, out.strides[0])
    return out


def fast_detect(img, thr):
    img = _c(img, np.uint8)
    cap = img.size
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast_detect(_p(img), img.shape[1], img.shape[0], img.strides[0], thr, _p(out), cap)
    return out[:n]


def fast_score_map(img):
    img = _c(img, np.uint8)
    out = np.zeros(img.shape, np.int32)
    lib().orc_fast_score_map(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(out))
    return out


def gaussian_blur(img):
    img = _c(img, np.uint8)
    out = np.zeros_like(img)
    lib().orc_gaussian_blur(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(out), out.strides[0])
    return out


def fast_atan2(y, x):
    return lib().orc_fast_atan2(y, x)


def sincos(r):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(r, C.byref(s), C.byref(c))
    return s.value, c.value


def orb_descriptor(img, x, y, angle):
    img = _c(img, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_orb_descriptor(_p(img), img.strides[0], x, y, angle, _p(d))
    return d


def hamming_swar(a, b):
    return lib().orc_hamming_swar(_p(_c(a, np.uint8)), _p(_c(b, np.uint8)))


def depth_to_float(d16, factor):
    d16 = _c(d16, np.uint16)
    out = np.zeros(d16.shape, np.float32)
    lib().orc_depth_to_float(_p(d16), d16.size, np.float32(factor), _p(out))
    return out


class FrameOracle:
    """Restatement of the parts of Frame the matchers read (reference include/Frame.h:183-299)."""

    def __init__(self, kps, desc, depth_f32, K4, bf, imw, imh, scale_factors, dist=None):
        self.L = lib()
        self.kps = _c(kps, KP_DTYPE)
        self.desc = _c(desc, np.uint8)
        self.N = len(self.kps)
        depth_f32 = _c(depth_f32, np.float32)
        K4 = _c(K4, np.float32)
        sf = _c(scale_factors, np.float32)
        if dist is None:
            self.h = self.L.orc_frame_create(_p(self.kps), _p(self.desc), self.N, _p(depth_f32), depth_f32.shape[1],
                                             depth_f32.shape[0], _p(K4), np.float32(bf), imw, imh, _p(sf), len(sf))
        else:   # Frame::UndistortKeyPoints + ComputeImageBounds with (k1, k2, p1, p2[, k3])
            d = _c(dist, np.float32)
            self.L.orc_frame_create_dist.restype = C.c_void_p
            self.L.orc_frame_create_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                     C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            self.h = self.L.orc_frame_create_dist(_p(self.kps), _p(self.desc), self.N, _p(depth_f32), depth_f32.shape[1],
                                                  depth_f32.shape[0], _p(K4), np.float32(bf), imw, imh, _p(sf), len(sf),
                                                  _p(d), len(d))
        self.uRight = np.zeros(self.N, np.float32)
        self.depth = np.zeros(self.N, np.float32)
        self.L.orc_frame_get_stereo(self.h, _p(self.uRight), _p(self.depth))

    def __del__(self):
        try:
            self.L.orc_frame_destroy(self.h)
        except Exception:
            pass

    def keys_un(self):
        out = np.zeros(max(self.N, 1), KP_DTYPE)
        self.L.orc_frame_get_keys_un.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_frame_get_keys_un(self.h, _p(out))
        return out[:self.N]

    def bounds(self):
        out = np.zeros(4, np.float32)
        self.L.orc_frame_get_bounds.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_frame_get_bounds(self.h, _p(out))
        return out

    def grid_csr(self):
        off = np.zeros(64 * 48 + 1, np.int32)
        idx = np.zeros(max(self.N, 1), np.int32)
        self.L.orc_frame_grid_csr(self.h, _p(off), _p(idx))
        return off, idx[:off[-1]]

    def features_in_area(self, x, y, r, minL=-1, maxL=-1):
        out = np.zeros(max(self.N, 1), np.int32)
        n = self.L.orc_frame_features_in_area(self.h, x, y, r, minL, maxL, _p(out), len(out))
        return out[:n]

    def unproject(self, Twc):
        Twc = _c(Twc, np.float32)
        w = np.zeros((self.N, 3), np.float32)
        v = np.zeros(self.N, np.uint8)
        self.L.orc_frame_unproject(self.h, _p(Twc), _p(w), _p(v))
        return w, v


FRUSTUM_POINT_DTYPE = np.dtype([("world", "<f4", (3,)), ("normal", "<f4", (3,)), ("min_distance", "<f4"),
                                ("max_distance", "<f4")])
FRUSTUM_LINE_DTYPE = np.dtype([("world", "<f8", (6,)), ("normal", "<f8", (3,)), ("min_distance", "<f4"),
                               ("max_distance", "<f4")])
FRUSTUM_OUT_DTYPE = np.dtype([("in_view", "<i4"), ("level", "<i4"), ("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"),
                              ("view_cos", "<f4")])
FRUSTUM_LINE_OUT_DTYPE = np.dtype([("in_view", "<i4"), ("level", "<i4"), ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"),
                                   ("y2", "<f4"), ("view_cos", "<f4")])


def logf(x):
    L = lib()
    L.orc_logf.restype = C.c_float
    L.orc_logf.argtypes = [C.c_float]
    return L.orc_logf(float(x))


def is_in_frustum(cam9, bf, Tcw, scale_factor, nlevels, pts, limit):
    """Frame::isInFrustum(MapPoint*, viewingCosLimit), src/Frame.cc:602-657, for an array of map points."""
    L = lib()
    p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
    out = np.zeros(len(p), FRUSTUM_OUT_DTYPE)
    L.orc_is_in_frustum.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                    C.c_void_p]
    L.orc_is_in_frustum(_p(_c(cam9, np.float32)), float(bf), _p(_c(Tcw, np.float32).reshape(16)),
                        logf(np.float32(scale_factor)), int(nlevels), _p(p), len(p), float(limit), _p(out))
    return out


def is_in_frustum_lines(cam9, Tcw, scale_factor, lines, limit):
    L = lib()
    l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
    out = np.zeros(len(l), FRUSTUM_LINE_OUT_DTYPE)
    L.orc_is_in_frustum_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.orc_is_in_frustum_lines(_p(_c(cam9, np.float32)), _p(_c(Tcw, np.float32).reshape(16)), logf(np.float32(scale_factor)),
                              _p(l), len(l), float(limit), _p(out))
    return out


def fuse_search(kf: "FrameOracle", Tcw, scale_factor, inv_level_sigma2, pts, descs, skip, th):
    """Search part of ORBmatcher::Fuse(pKF, vpMapPoints, th): returns (best_idx, best_dist) per map point."""
    L = lib()
    p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
    bi = np.zeros(len(p), np.int32)
    bd = np.zeros(len(p), np.int32)
    inv = _c(inv_level_sigma2, np.float32)
    L.orc_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_fuse_search(kf.h, _p(_c(Tcw, np.float32).reshape(16)), _p(inv), logf(np.float32(scale_factor)), len(inv), _p(p),
                      _p(_c(descs, np.uint8)), _p(_c(skip, np.uint8)), len(p), float(th), _p(bi), _p(bd))
    return bi, bd


def lsd_fuse_search(cam9, Tcw, scale_factor, scale, lines, descs, skip, kf_keylines, kf_desc, th):
    """Search part of LSDmatcher::Fuse(pKF, vpMapLines, th): returns (best_idx, best_dist) per map line."""
    L = lib()
    l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
    sc = _c(scale, np.float32)
    kf = _line_recs(kf_keylines)
    bi = np.zeros(len(l), np.int32)
    bd = np.zeros(len(l), np.int32)
    L.orc_lsd_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_lsd_fuse_search(_p(_c(cam9, np.float32)), _p(_c(Tcw, np.float32).reshape(16)), logf(np.float32(scale_factor)), _p(sc), len(sc),
                          _p(l), _p(_c(descs, np.uint8)), _p(_c(skip, np.uint8)), len(l), _p(kf), _p(_c(kf_desc, np.uint8)), len(kf),
                          float(th), _p(bi), _p(bd))
    return bi, bd


def lsd_fuse_search_sim3(cam9, Scw, scale_factor, scale, lines, descs, skip, kf_keylines, kf_desc, th):
    """Search part of LSDmatcher::Fuse(pKF, Scw, vpLines, th, vpReplaceLine): returns (best_idx, best_dist) per map line."""
    L = lib()
    l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
    sc = _c(scale, np.float32)
    kf = _line_recs(kf_keylines)
    bi = np.zeros(len(l), np.int32)
    bd = np.zeros(len(l), np.int32)
    L.orc_lsd_fuse_search_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_lsd_fuse_search_sim3(_p(_c(cam9, np.float32)), _p(_c(Scw, np.float32).reshape(16)), logf(np.float32(scale_factor)), _p(sc),
                               len(sc), _p(l), _p(_c(descs, np.uint8)), _p(_c(skip, np.uint8)), len(l), _p(kf),
                               _p(_c(kf_desc, np.uint8)), len(kf), float(th), _p(bi), _p(bd))
    return bi, bd


def lsd_search_by_projection_kf(cam9, Scw, scale_factor, scale, lines, descs, skip, kf_keylines, kf_desc, matched, th):
    """LSDmatcher::SearchByProjection(pKF, Scw, vpLines, vpMatched, th): returns (nmatches, new_match per key line)."""
    L = lib()
    l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
    sc = _c(scale, np.float32)
    kf = _line_recs(kf_keylines)
    out = np.full(len(kf), -1, np.int32)
    L.orc_lsd_search_by_projection_kf.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    n = L.orc_lsd_search_by_projection_kf(_p(_c(cam9, np.float32)), _p(_c(Scw, np.float32).reshape(16)), logf(np.float32(scale_factor)),
                                          _p(sc), len(sc), _p(l), _p(_c(descs, np.uint8)), _p(_c(skip, np.uint8)), len(l), _p(kf),
                                          _p(_c(kf_desc, np.uint8)), len(kf), _p(_c(matched, np.uint8)), int(th), _p(out))
    return n, out


def lsd_search_by_sim3(cam9, T1w, T2w, s12, R12, t12, scale_factor, scale, lines1, descs1, skip1, kf1_keylines, kf1_desc,
                       lines2, descs2, skip2, kf2_keylines, kf2_desc, th):
    """LSDmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th): returns (nFound, out12[i1] = i2 or -1)."""
    L = lib()
    l1, l2 = np.ascontiguousarray(lines1, FRUSTUM_LINE_DTYPE), np.ascontiguousarray(lines2, FRUSTUM_LINE_DTYPE)
    sc = _c(scale, np.float32)
    k1, k2 = _line_recs(kf1_keylines), _line_recs(kf2_keylines)
    assert len(l1) == len(k1) and len(l2) == len(k2)
    out = np.full(len(l1), -1, np.int32)
    L.orc_lsd_search_by_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    f = lambda a, n: _p(_c(a, np.float32).reshape(n))
    n = L.orc_lsd_search_by_sim3(_p(_c(cam9, np.float32)), f(T1w, 16), f(T2w, 16), float(s12), f(R12, 9), f(t12, 3),
                                 logf(np.float32(scale_factor)), _p(sc), len(sc), _p(l1), _p(_c(descs1, np.uint8)),
                                 _p(_c(skip1, np.uint8)), _p(k1), _p(_c(kf1_desc, np.uint8)), len(l1), _p(l2),
                                 _p(_c(descs2, np.uint8)), _p(_c(skip2, np.uint8)), _p(k2), _p(_c(kf2_desc, np.uint8)), len(l2),
                                 float(th), _p(out))
    return n, out


def search_by_projection_kf(kf: "FrameOracle", Scw, scale_factor, nlevels, pts, descs, skip, matched, th):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th): returns (nmatches, new_match per keypoint)."""
    L = lib()
    p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
    m = _c(matched, np.uint8)
    out = np.full(len(m), -1, np.int32)
    L.orc_search_by_projection_kf.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_float, C.c_void_p]
    n = L.orc_search_by_projection_kf(kf.h, _p(_c(Scw, np.float32).reshape(16)), logf(np.float32(scale_factor)), int(nlevels), _p(p),
                                      _p(_c(descs, np.uint8)), _p(_c(skip, np.uint8)), len(p), _p(m), float(th), _p(out))
    return n, out


def search_by_projection_reloc(cur: "FrameOracle", Tcw, scale_factor, nlevels, pts, descs, kf_angles, skip, matched, th, orb_dist,
                               check_ori=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): (nmatches, new_match per keypoint)."""
    L = lib()
    p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
    m = _c(matched, np.uint8)
    out = np.full(len(m), -1, np.int32)
    L.orc_search_by_projection_reloc.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]
    n = L.orc_search_by_projection_reloc(cur.h, _p(_c(Tcw, np.float32).reshape(16)), logf(np.float32(scale_factor)), int(nlevels), _p(p),
                                         _p(_c(descs, np.uint8)), _p(_c(kf_angles, np.float32)), _p(_c(skip, np.uint8)), len(p), _p(m),
                                         float(th), int(orb_dist), int(bool(check_ori)), _p(out))
    return n, out


def search_for_initialization(f1: "FrameOracle", f2: "FrameOracle", prev_matched, window_size=100, nnratio=0.9, check_ori=True):
    """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:409-524:
    returns (nmatches, matches12 [F1.N], prev_matched after the update [F1.N, 2])."""
    L = lib()
    pm = _c(prev_matched, np.float32).reshape(-1, 2).copy()
    out = np.full(len(pm), -1, np.int32)
    L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    n = L.orc_search_for_initialization(f1.h, f2.h, _p(pm), int(window_size), float(nnratio), int(bool(check_ori)), _p(out))
    return n, out, pm


def search_by_sim3(kf1: "FrameOracle", kf2: "FrameOracle", T1w, T2w, s12, R12, t12, scale_factor, nlevels, pts1, descs1, skip1,
                   pts2, descs2, skip2, th):
    """ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th): returns (nFound, out12[i1] = i2 or -1)."""
    L = lib()
    p1, p2 = np.ascontiguousarray(pts1, FRUSTUM_POINT_DTYPE), np.ascontiguousarray(pts2, FRUSTUM_POINT_DTYPE)
    out = np.full(len(p1), -1, np.int32)
    L.orc_search_by_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                     C.c_void_p]
    n = L.orc_search_by_sim3(kf1.h, kf2.h, _p(_c(T1w, np.float32).reshape(16)), _p(_c(T2w, np.float32).reshape(16)), float(s12),
                             _p(_c(R12, np.float32).reshape(9)), _p(_c(t12, np.float32).reshape(3)), logf(np.float32(scale_factor)),
                             int(nlevels), _p(p1), _p(_c(descs1, np.uint8)), _p(_c(skip1, np.uint8)), _p(p2), _p(_c(descs2, np.uint8)),
                             _p(_c(skip2, np.uint8)), float(th), _p(out))
    return n, out


def fuse_search_sim3(kf: "FrameOracle", Scw, scale_factor, inv_level_sigma2, pts, descs, skip, th):
    """Search part of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (similarity pose, no chi-square gate)."""
    L = lib()
    p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
    bi = np.zeros(len(p), np.int32)
    bd = np.zeros(len(p), np.int32)
    inv = _c(inv_level_sigma2, np.float32)
    L.orc_fuse_search_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_fuse_search_sim3(kf.h, _p(_c(Scw, np.float32).reshape(16)), _p(inv), logf(np.float32(scale_factor)), len(inv), _p(p),
                           _p(_c(descs, np.uint8)), _p(_c(skip, np.uint8)), len(p), float(th), _p(bi), _p(bd))
    return bi, bd


def undistort_points(xy, K4, dist):
    """cv::undistortPoints(xy, K, dist, Mat(), K), float32 in/out (N x 2)."""
    xy = _c(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    d = _c(dist, np.float32)
    lib().orc_undistort_points(_p(xy), len(xy), _p(_c(K4, np.float32)), _p(d), len(d), _p(out))
    return out


def image_bounds(cols, rows, K4, dist):
    out = np.zeros(4, np.float32)
    d = _c(dist, np.float32)
    lib().orc_image_bounds(int(cols), int(rows), _p(_c(K4, np.float32)), _p(d), len(d), _p(out))
    return out


def search_by_projection_last(cur: FrameOracle, last: FrameOracle, Tcw_cur, Tcw_last, last_mp, th=15.0,
                              mono=False, check_ori=True, cur_mp=None, cur_obs=None):
    last_mp = _c(last_mp, MAPPOINT_DTYPE)
    out = np.full(cur.N, -1, np.int32) if cur_mp is None else _c(cur_mp, np.int32).copy()
    n = lib().orc_search_by_projection_last(cur.h, last.h, _p(_c(Tcw_cur, np.float32)), _p(_c(Tcw_last, np.float32)),
                                            _p(last_mp), th, int(mono), int(check_ori),
                                            _p(None if cur_obs is None else _c(cur_obs, np.uint8)), _p(out))
    return n, out


def search_by_projection_map(frame: FrameOracle, mps, th, nnratio, frame_mp=None, claim_obs=None):
    mps = _c(mps, TRACKED_DTYPE)
    out = np.full(frame.N, -1, np.int32) if frame_mp is None else _c(frame_mp, np.int32).copy()
    n = lib().orc_search_by_projection_map(frame.h, _p(mps), len(mps), th, nnratio,
                                           _p(None if claim_obs is None else _c(claim_obs, np.uint8)), _p(out))
    return n, out


def bf_knn(Q, T, k):
    Q, T = _c(Q, np.uint8), _c(T, np.uint8)
    idx = np.zeros((len(Q), k), np.int32)
    dist = np.zeros((len(Q), k), np.int32)
    lib().orc_bf_knn_hamming(_p(Q), len(Q), _p(T), len(T), k, _p(idx), _p(dist))
    return idx, dist


def lsd_search_by_descriptor(desc_kf, kf_has_line, desc_f):
    """LSDmatcher::SearchByDescriptor(pKF, currentF, matches): out[frame line] = KF line or -1."""
    dk, df = _c(desc_kf, np.uint8), _c(desc_f, np.uint8)
    out = np.full(len(df), -1, np.int32)
    L = lib()
    L.orc_lsd_search_by_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    n = L.orc_lsd_search_by_descriptor(_p(dk), len(dk), _p(_c(kf_has_line, np.uint8)), _p(df), len(df), _p(out))
    return n, out


def lsd_search_for_triangulation(desc1, desc2, has1, has2):
    """LSDmatcher::SearchForTriangulation(pKF1, pKF2, pairs): out12[line of KF1] = line of KF2 or -1."""
    d1, d2 = _c(desc1, np.uint8), _c(desc2, np.uint8)
    out = np.full(len(d1), -1, np.int32)
    L = lib()
    L.orc_lsd_search_for_triangulation.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.orc_lsd_search_for_triangulation(_p(d1), len(d1), _p(d2), len(d2), _p(_c(has1, np.uint8)), _p(_c(has2, np.uint8)), _p(out))
    return n, out


def lsd_search_by_gap(desc_q, desc_t, train_has_line=None):
    """LSDmatcher::SearchByDescriptor(pKF, pKF2, ...) / SerachForInitialize: out[query line] = train line or -1."""
    dq, dt = _c(desc_q, np.uint8), _c(desc_t, np.uint8)
    out = np.full(len(dq), -1, np.int32)
    L = lib()
    L.orc_lsd_search_by_gap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    has = None if train_has_line is None else _c(train_has_line, np.uint8)
    n = L.orc_lsd_search_by_gap(_p(dq), len(dq), _p(dt), len(dt), _p(has), _p(out))
    return n, out


LINE_DTYPE = np.dtype([("pt_x", "<f4"), ("pt_y", "<f4"), ("angle", "<f4"), ("octave", "<i4")])          # orc::LineRec
MAPLINE_DTYPE = np.dtype([("valid", "<i4"), ("octave", "<i4"), ("obs_positive", "<i4"), ("pad", "<i4"),
                          ("world", "<f8", (6,)), ("desc", "u1", (32,))])                                 # orc::MapLineRec
TRACKED_LINE_DTYPE = np.dtype([("in_view", "<i4"), ("level", "<i4"), ("obs_positive", "<i4"), ("x1", "<f4"),
                               ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("view_cos", "<f4"),
                               ("desc", "u1", (32,))])                                                     # orc::TrackedLineRec


def _line_recs(keylines):
    out = np.zeros(len(keylines), LINE_DTYPE)
    for f in ("pt_x", "pt_y", "angle", "octave"):
        out[f] = keylines[f]
    return out


def lsd_search_by_projection_last(cam9, Tcw_cur, Tcw_last, scale, last_lines, cur_keylines, cur_desc, th, mono, nnratio,
                                  cur_ml, cur_obs=None):
    """LSDmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/LSDmatcher.cpp:20-139.
    cam9 = (fx, fy, cx, cy, mb, minX, maxX, minY, maxY)."""
    L = lib()
    assert L.orc_sizeof_maplinerec() == MAPLINE_DTYPE.itemsize
    cam = _c(cam9, np.float32)
    tc, tl = _c(Tcw_cur, np.float32).reshape(16), _c(Tcw_last, np.float32).reshape(16)
    sc = _c(scale, np.float32)
    ll = np.ascontiguousarray(last_lines, MAPLINE_DTYPE)
    cur = _line_recs(cur_keylines)
    cd = _c(cur_desc, np.uint8)
    out = _c(cur_ml, np.int32).copy()
    obs = None if cur_obs is None else _c(cur_obs, np.uint8)
    L.orc_lsd_search_by_projection_last.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                                                       C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    n = L.orc_lsd_search_by_projection_last(_p(cam), _p(tc), _p(tl), _p(sc), _p(ll), len(ll), _p(cur), _p(cd), len(cur),
                                            th, int(mono), nnratio, _p(obs), _p(out))
    return n, out


def lsd_search_by_projection_map(scale, lines, cur_keylines, cur_desc, th, nnratio, cur_ml, cur_obs=None):
    """LSDmatcher::SearchByProjection(F, vpMapLines, th), src/LSDmatcher.cpp:141-211."""
    L = lib()
    assert L.orc_sizeof_trackedlinerec() == TRACKED_LINE_DTYPE.itemsize
    sc = _c(scale, np.float32)
    tl = np.ascontiguousarray(lines, TRACKED_LINE_DTYPE)
    cur = _line_recs(cur_keylines)
    cd = _c(cur_desc, np.uint8)
    out = _c(cur_ml, np.int32).copy()
    obs = None if cur_obs is None else _c(cur_obs, np.uint8)
    L.orc_lsd_search_by_projection_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                                   C.c_float, C.c_void_p, C.c_void_p]
    n = L.orc_lsd_search_by_projection_map(_p(sc), _p(tl), len(tl), _p(cur), _p(cd), len(cur), th, nnratio, _p(obs), _p(out))
    return n, out


def match_orb_points(cur_desc, last_desc, last_mp, last_outlier):
    cd, ld = _c(cur_desc, np.uint8), _c(last_desc, np.uint8)
    out = np.full(len(cd), -1, np.int32)
    n = lib().orc_match_orb_points(_p(cd), len(cd), _p(ld), len(ld), _p(_c(last_mp, np.int32)),
                                   _p(_c(last_outlier, np.uint8)), _p(out))
    return n, out


def eig33sym(K):
    """LA::eig33sym (Eigen SelfAdjointEigenSolver<Matrix3d>): ascending eigenvalues, eigenvectors in columns."""
    K = _c(K, np.float64)
    s = np.zeros(3)
    V = np.zeros((3, 3))
    lib().orc_eig33sym(_p(K), _p(s), _p(V))
    return s, V


def ahc_planes(depth16, K4, depthfactor):
    """PlaneDetection::readDepthImage + runPlaneDetection (reference src/PlaneExtractor.cpp:28-63).
    Returns dict(planes=[n,8] normal|center|mse|curvature, N, rid, seg=[h,w] uint8, members=[...],
    blocks=[nb,17], block_valid, block_N)."""
    d = _c(depth16, np.uint16)
    h, w = d.shape
    L = lib()
    H = L.orc_ahc_run(_p(d), w, h, _p(_c(K4, np.float32)), np.float32(depthfactor))
    if not H:
        raise RuntimeError(L.orc_last_error().decode())
    try:
        n, nb = L.orc_ahc_num_planes(H), L.orc_ahc_num_blocks(H)
        planes = np.zeros((n, 8))
        nrid = np.zeros((n, 2), np.int32)
        L.orc_ahc_get_planes(H, _p(planes), _p(nrid))
        seg = np.zeros((h, w), np.uint8)
        L.orc_ahc_get_seg(H, _p(seg))
        members = []
        for i in range(n):
            m = np.zeros(L.orc_ahc_member_count(H, i), np.int32)
            L.orc_ahc_get_members(H, i, _p(m))
            members.append(m)
        blocks = np.zeros((nb, 17))
        vn = np.zeros((nb, 2), np.int32)
        L.orc_ahc_get_blocks(H, _p(blocks), _p(vn))
    finally:
        L.orc_ahc_free(H)
    return dict(planes=planes, N=nrid[:, 0], rid=nrid[:, 1], seg=seg, members=members, blocks=blocks,
                block_valid=vn[:, 0], block_N=vn[:, 1])


KEYLINE_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("ptX", "<f4"), ("ptY", "<f4"),
                          ("response", "<f4"), ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"),
                          ("endPointX", "<f4"), ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"),
                          ("sPointInOctaveY", "<f4"), ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                          ("lineLength", "<f4"), ("numOfPixels", "<i4")])


def extract_lines(gray, max_lines=40, stages=False, rect_mode=0, trace=False):
    """LineSegment::ExtractLineSegment (reference src/LSDextractor.cpp:12-43): LSD detect, keep the 40
    highest-response lines, LBD descriptors, normalised line equations.
    rect_mode: lsd.cpp's reading - 0 the OpenCV 3.4 source text (integer corners, `double(n) + 1` in nfa(); default), 1 the LSD paper's
    (real-valued corners, log_gamma(n + 1): rounds 2-3), 2 integer corners with log_gamma(n + 1) (round 4).
    Returns dict(lines (KEYLINE_DTYPE), desc [n,32], descf [n,72], lineF [n,3], detected[, stage images]
    [, rect_counts [calls,2] int32 in call order, segments [n,4] float32 before the key-line stage, seg_width / seg_prec /
    seg_nfa: what cv::LineSegmentDetector::detect reports beside the segments])."""
    g = _c(gray, np.uint8)
    h, w = g.shape
    L = lib()
    L.orc_lines_run_mode.restype = C.c_void_p
    L.orc_lines_run_mode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_lines_trace_info.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_lines_trace_get.argtypes = [C.c_void_p] * 3
    L.orc_lines_free.argtypes = [C.c_void_p]
    L.orc_lines_info.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_lines_get.argtypes = [C.c_void_p] * 5
    L.orc_lines_get_stages.argtypes = [C.c_void_p] * 6
    assert L.orc_sizeof_keyline() == KEYLINE_DTYPE.itemsize
    H = L.orc_lines_run_mode(_p(g), w, h, max_lines, int(rect_mode))
    if not H:
        raise RuntimeError(L.orc_last_error().decode())
    try:
        info = np.zeros(4, np.int32)
        L.orc_lines_info(H, _p(info))
        n, detected, sw, sh = (int(v) for v in info)
        kl = np.zeros(n, KEYLINE_DTYPE)
        desc = np.zeros((n, 32), np.uint8)
        descf = np.zeros((n, 72), np.float32)
        lineF = np.zeros((n, 3))
        L.orc_lines_get(H, _p(kl), _p(desc), _p(descf), _p(lineF))
        out = dict(lines=kl, desc=desc, descf=descf, lineF=lineF, detected=detected)
        if stages:
            scaled = np.zeros((sh, sw), np.uint8)
            modgrad, angles = np.zeros((sh, sw)), np.zeros((sh, sw))
            gx, gy = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
            L.orc_lines_get_stages(H, _p(scaled), _p(modgrad), _p(angles), _p(gx), _p(gy))
            out.update(scaled=scaled, modgrad=modgrad, angles=angles, gx=gx, gy=gy)
        if trace:
            tn = np.zeros(2, np.int32)
            L.orc_lines_trace_info(H, _p(tn))
            rc, sg = np.zeros((int(tn[0]), 2), np.int32), np.zeros((int(tn[1]), 4), np.float32)
            L.orc_lines_trace_get(H, _p(rc), _p(sg))
            L.orc_lines_trace_get_info.argtypes = [C.c_void_p, C.c_void_p]
            si = np.zeros((int(tn[1]), 3))
            L.orc_lines_trace_get_info(H, _p(si))
            out.update(rect_counts=rc, segments=sg, seg_width=si[:, 0], seg_prec=si[:, 1] * np.pi, seg_nfa=si[:, 2])
    finally:
        L.orc_lines_free(H)
    return out


class VocabularyOracle:
    """DBoW2 TemplatedVocabulary<FORB>: text loader + transform (reference Thirdparty/DBoW2)."""

    def __init__(self, text: str):
        self.L_ = lib()
        self.h = self.L_.orc_voc_create(text.encode())
        if not self.h:
            raise RuntimeError(self.L_.orc_last_error().decode())
        info = np.zeros(5, np.int32)
        self.L_.orc_voc_info(self.h, _p(info))
        self.k, self.L, self.scoring, self.weighting, self.n_nodes = (int(v) for v in info)

    def __del__(self):
        try:
            self.L_.orc_voc_free(self.h)
        except Exception:
            pass

    def nodes(self):
        n = self.n_nodes
        parent, word = np.zeros(n, np.int32), np.zeros(n, np.int32)
        desc, weight = np.zeros((n, 32), np.uint8), np.zeros(n)
        self.L_.orc_voc_get_nodes(self.h, _p(parent), _p(word), _p(desc), _p(weight))
        return parent, word, desc, weight

    def transform_each(self, desc, levelsup=4):
        desc = _c(desc, np.uint8)
        n = len(desc)
        word, nid, weight = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
        self.L_.orc_voc_transform_each(self.h, _p(desc), n, levelsup, _p(word), _p(weight), _p(nid))
        return word, weight, nid

    def bow_vector(self, desc, levelsup=4):
        desc = _c(desc, np.uint8)
        cap = max(len(desc), 1)
        ids, vals = np.zeros(cap, np.int32), np.zeros(cap)
        n = self.L_.orc_voc_transform_bow(self.h, _p(desc), len(desc), levelsup, _p(ids), _p(vals), cap)
        return ids[:n], vals[:n]


def search_by_bow(nid_kf, nid_f, desc_kf, angle_kf, kf_mp, desc_f, angle_f, nnratio, check_ori=True):
    """ORBmatcher::SearchByBoW(pKF, F, matches). nid_*: FeatureVector node per feature (-1 = stopped word)."""
    nid_kf, nid_f = _c(nid_kf, np.int32), _c(nid_f, np.int32)
    out = np.full(len(nid_f), -1, np.int32)
    n = lib().orc_search_by_bow(_p(nid_kf), len(nid_kf), _p(nid_f), len(nid_f), _p(_c(desc_kf, np.uint8)),
                                _p(_c(angle_kf, np.float32)), _p(_c(kf_mp, np.int32)), _p(_c(desc_f, np.uint8)),
                                _p(_c(angle_f, np.float32)), np.float32(nnratio), int(check_ori), _p(out))
    return n, out


def search_by_bow_kf(nid1, nid2, desc1, angle1, mp1, desc2, angle2, mp2, nnratio, check_ori=True):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12): returns (nmatches, out2) with out2[keypoint of KF2] = keypoint
    of KF1 or -1 (vpMatches12[idx1] = vpMapPoints2[idx2])."""
    nid1, nid2 = _c(nid1, np.int32), _c(nid2, np.int32)
    out = np.full(len(nid2), -1, np.int32)
    L = lib()
    L.orc_search_by_bow_kf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    n = L.orc_search_by_bow_kf(_p(nid1), len(nid1), _p(nid2), len(nid2), _p(_c(desc1, np.uint8)), _p(_c(angle1, np.float32)),
                               _p(_c(mp1, np.int32)), _p(_c(desc2, np.uint8)), _p(_c(angle2, np.float32)), _p(_c(mp2, np.int32)),
                               float(nnratio), int(check_ori), _p(out))
    return n, out


def search_for_triangulation(kf1, kf2, F12, ex, ey, scale_factors, level_sigma2, only_stereo=False, check_ori=True):
    """ORBmatcher::SearchForTriangulation.  kf = dict(x, y, angle, u_right, octave, mp, nid, desc) per keypoint
    (mp >= 0: has a map point; nid < 0: stopped word).  Returns (nmatches, matches12)."""
    def pack(k):
        kp = np.stack([_c(k["x"], np.float32), _c(k["y"], np.float32), _c(k["angle"], np.float32),
                       _c(k["u_right"], np.float32)], 1).astype(np.float32)
        io = np.stack([_c(k["octave"], np.int32), _c(k["mp"], np.int32), _c(k["nid"], np.int32)], 1).astype(np.int32)
        return np.ascontiguousarray(kp), np.ascontiguousarray(io), _c(k["desc"], np.uint8)
    kp1, io1, d1 = pack(kf1)
    kp2, io2, d2 = pack(kf2)
    out = np.full(len(kp1), -1, np.int32)
    L = lib()
    L.orc_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int, C.c_void_p]
    n = L.orc_search_for_triangulation(_p(kp1), _p(io1), _p(d1), len(kp1), _p(kp2), _p(io2), _p(d2), len(kp2),
                                       _p(_c(F12, np.float32).reshape(9)), float(ex), float(ey), _p(_c(scale_factors, np.float32)),
                                       _p(_c(level_sigma2, np.float32)), int(only_stereo), int(check_ori), _p(out))
    return n, out


def cape_planes(depth_m, K4, patch=20, cos_angle_max=None, max_merge_dist=50.0):
    """PlaneDetection_CAPE::readDepthImage + runPlaneDetection (reference src/PlaneExtractor.cpp:102-191).
    depth_m: float32 metres (what Frame::ComputePlanes_CAPE passes).  Returns dict(planes=[n,7]
    normal|mean|d, MSE, score, nr_pts, seg, cells=[nc,16], cell_mst=[nc,3], cell_planar, cell_npts)."""
    d = _c(depth_m, np.float32)
    h, w = d.shape
    if cos_angle_max is None:
        cos_angle_max = np.float32(np.cos(np.pi / 12))     # include/PlaneExtractor.h:111
    L = lib()
    H = L.orc_cape_run(_p(d), w, h, _p(_c(K4, np.float32)), patch, np.float32(cos_angle_max), np.float32(max_merge_dist))
    if not H:
        raise RuntimeError(L.orc_last_error().decode())
    try:
        n, nc = L.orc_cape_num_planes(H), L.orc_cape_num_cells(H)
        planes = np.zeros((n, 7))
        ms = np.zeros((n, 2), np.float32)
        npts = np.zeros(n, np.int32)
        L.orc_cape_get_planes(H, _p(planes), _p(ms), _p(npts))
        seg = np.zeros((h, w), np.uint8)
        L.orc_cape_get_seg(H, _p(seg))
        cells = np.zeros((nc, 16))
        mst = np.zeros((nc, 3), np.float32)
        pn = np.zeros((nc, 2), np.int32)
        L.orc_cape_get_cells(H, _p(cells), _p(mst), _p(pn))
    finally:
        L.orc_cape_free(H)
    return dict(planes=planes, MSE=ms[:, 0], score=ms[:, 1], nr_pts=npts, seg=seg, cells=cells, cell_mst=mst,
                cell_planar=pn[:, 0], cell_npts=pn[:, 1])


# ---------------------------------------------------------------------------------------------------
# Frame::ComputePlanes after the extractor: voxel grid, RANSAC refit, surface normals (oracle/post_oracle.cpp)

def post_voxel_grid(xyz, leaf=0.05):
    L = lib()
    p = _c(xyz, np.float32).reshape(-1, 3)
    out = np.zeros_like(p)
    L.orc_post_voxel_grid.restype = C.c_int
    L.orc_post_voxel_grid.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    n = L.orc_post_voxel_grid(_p(p), len(p), np.float32(leaf), _p(out))
    return out[:n].copy()


def post_refit(coef4, xyz, dist_threshold):
    """Frame::MaxPointDistanceFromPlane -> (valid, coef)."""
    L = lib()
    c = _c(coef4, np.float32).copy()
    p = _c(xyz, np.float32).reshape(-1, 3)
    L.orc_post_refit.restype = C.c_int
    L.orc_post_refit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double]
    v = L.orc_post_refit(_p(c), _p(p), len(p), float(dist_threshold))
    return bool(v), c


def post_surface_normals(depth_m, K4, max_point_dist):
    """-> (cloud [H3, W3, 3], normals [H3, W3, 3]) of the 3x-subsampled organized cloud."""
    L = lib()
    d = _c(depth_m, np.float32)
    h, w = d.shape
    W, H = (w + 2) // 3, (h + 2) // 3
    cloud = np.zeros((H, W, 3), np.float32)
    nrm = np.zeros((H, W, 3), np.float32)
    L.orc_post_surface_normals.restype = None
    L.orc_post_surface_normals.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long] + [C.c_float] * 5 + [C.c_void_p] * 3
    L.orc_post_surface_normals(_p(d), w, h, w, *(np.float32(v) for v in K4), np.float32(max_point_dist), _p(cloud), _p(nrm), None)
    return cloud, nrm


def post_surface_normal_records(cloud, nrm):
    """The vSurfaceNormal list (src/Frame.cc:1069-1090): odd rows, odd columns."""
    H, W, _ = cloud.shape
    rows, cols = np.arange(1, H, 2), np.arange(1, W, 2)
    rr, cc = np.meshgrid(rows, cols, indexing="ij")
    return nrm[rr, cc].reshape(-1, 3), cloud[rr, cc].reshape(-1, 3), (cc * 3).reshape(-1), (rr * 3).reshape(-1)


def ahc_post_planes(depth16, K4, depthfactor, ahc, max_point_dist, dist_threshold):
    """The per-plane loop of Frame::ComputePlanes (src/Frame.cc:952-1011) on the oracle's AHC output `ahc`.
    Returns list of dict(coef, accepted, voxels) per extracted plane and plane_num."""
    d = _c(depth16, np.uint16)
    h, w = d.shape
    fx, fy, cx, cy = (np.float32(v) for v in K4)
    out, fail = [], 0
    for i, mem in enumerate(ahc["members"]):
        j = np.asarray(mem, np.int64)
        row, col = j // w, j % w
        z = d[row, col].astype(np.float64) * np.float64(np.float32(depthfactor))
        far = z > 5.0
        x = np.where(far, 0.0, (col.astype(np.float64) - np.float64(cx)) * z / np.float64(fx))
        y = np.where(far, 0.0, (row.astype(np.float64) - np.float64(cy)) * z / np.float64(fy))
        z = np.where(far, 0.0, z)
        pts = np.stack([x, y, z], 1).astype(np.float32)
        pts = pts[~(pts[:, 2] > np.float32(max_point_dist))]
        nrm, cen = ahc["planes"][i, 0:3], ahc["planes"][i, 3:6]
        dd = np.float32(-(nrm[0] * cen[0] + nrm[1] * cen[1] + nrm[2] * cen[2]))
        coef = np.array([nrm[0], nrm[1], nrm[2], dd], np.float32)
        vox = post_voxel_grid(pts)
        rec = dict(coef=coef, accepted=False, voxels=vox)
        if dd > np.float32(max_point_dist) or len(vox) < 100:
            fail += 1
        else:
            ok, c2 = post_refit(coef, vox, dist_threshold)
            if ok:
                rec["coef"], rec["accepted"] = c2, True
        out.append(rec)
    return out, len(ahc["members"]) - fail


def cape_post_planes(depth_m, K4, cape, max_point_dist, dist_threshold):
    """The per-plane loop of Frame::ComputePlanes_CAPE (src/Frame.cc:1111-1141) on the oracle's CAPE output."""
    d = _c(depth_m, np.float32)
    h, w = d.shape
    fx, fy, cx, cy = (np.float64(np.float32(v)) for v in K4)
    seg = cape["seg"]
    out, fail = [], 0
    jj, ii = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    z = d.astype(np.float64)
    cloud = np.stack([(jj - cx) * z / fx, (ii - cy) * z / fy, z], -1).astype(np.float32)
    for i in range(len(cape["planes"])):
        pts = cloud[seg == i + 1]
        P = cape["planes"][i]
        coef = np.array([P[0], P[1], P[2], P[6]], np.float32)
        vox = post_voxel_grid(pts)
        rec = dict(coef=coef, accepted=False, voxels=vox)
        if P[6] > np.float64(np.float32(max_point_dist)) or len(vox) < 100:
            fail += 1
        else:
            ok, c2 = post_refit(coef, vox, dist_threshold)
            if ok:
                rec["coef"], rec["accepted"] = c2, True
            else:
                fail += 1
        out.append(rec)
    return out, len(cape["planes"]) - fail
