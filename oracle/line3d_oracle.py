"""oracle/line3d_oracle.py — TEST INFRASTRUCTURE (see oracle.h): numpy restatement of Frame::isLineGood
(reference src/Frame.cc:481-558) and of the helpers it calls in src/LineExtractor.cpp (depthStdDev :1180,
compPt3dCov :1196, extract3dline_mahdist :1266, verify3dLine :1362, mah_dist3d_pt_line :1419, computeLine3d_svd
:1157, projectPt3d2Ln3d :278, random_unique include/LSDextractor.h:241).  Only tests/ may import it.

Parity status: the shipped behaviour (mK is CV_32F but read with at<double>) is reproduced exactly — it is a NaN
cascade that rejects every line, see `focal_as_reference_reads_it`; the intended behaviour (f = fx) uses
numpy.linalg.svd where the reference uses cv::SVD, so it is unpinned beyond rounding.  rand() is glibc's TYPE_3
generator, pinned against the C library itself in tests/test_oracle_cpu2.py."""
import numpy as np


class GlibcRand:
    """rand() of glibc after srand(seed): r[i] = r[i-31] + r[i-3] (mod 2^32), output >> 1, 310 outputs discarded."""

    def __init__(self, seed=1):
        r = [0] * 344
        r[0] = seed if seed else 1
        for i in range(1, 31):
            r[i] = (16807 * r[i - 1]) % 2147483647
        for i in range(31, 34):
            r[i] = r[i - 31]
        for i in range(34, 344):
            r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
        self.r = r

    def __call__(self):
        v = (self.r[-31] + self.r[-3]) & 0xFFFFFFFF
        self.r.append(v)
        return v >> 1


def focal_as_reference_reads_it(K32):
    """K.at<double>(0,0) on the CV_32F mK: the bytes of (fx, 0.0f) as one double — a subnormal ~5.6e-315."""
    return float(np.frombuffer(np.ascontiguousarray(K32, np.float32).tobytes()[:8], np.float64)[0])


def _depth_std_dev(d):
    return 0.00273 * d * d + 0.00074 * d + (-0.00058)


def _comp_pt3d_cov(p, f):
    with np.errstate(all="ignore"):
        J = np.array([[p[2] / f, 0, p[0] / p[2]], [0, p[2] / f, p[1] / p[2]], [0, 0, 1.0]])
        C = np.diag([1.0, 1.0, _depth_std_dev(p[2]) ** 2])
        cov = (J @ C) @ J.T                      # inf * 0 = NaN exactly as in cv::gemm
    if not np.isfinite(cov).all():
        return np.full((3, 3), np.nan)           # cv::SVD of it: NaN singular values -> DU = NaN
    U, w, _ = np.linalg.svd(cov)
    return np.diag(1.0 / np.sqrt(w)) @ U.T


def _mah_dist(pos, DU, q1, q2):
    with np.errstate(all="ignore"):
        a = DU @ (pos - q1)
        b = DU @ (pos - q2)
        num = np.cross(a, b)
        den = a - b
        return np.sqrt((num @ num) / (den @ den))


def _project(P, mid, drct):
    return mid + drct * ((drct @ (P - mid)) / (drct @ drct))


def _verify(pts, A, B):
    v = (pts - A) @ (B - A)
    i1, i2 = 0, 0
    minv, maxv = 100.0, -100.0
    for i, x in enumerate(v):
        if x < minv:
            minv, i1 = x, i
        if x > maxv:
            maxv, i2 = x, i
    C = _project(pts[i1], (A + B) * 0.5, B - A)
    D = _project(pts[i2], (A + B) * 0.5, B - A)
    cd = np.linalg.norm(D - C)
    if cd < 1e-10:
        return False
    cells = np.zeros(10, int)
    for X in pts:
        lam = abs((X - C) @ (D - C) / cd / cd)
        cells[9 if lam >= 1 else int(np.floor(lam * 10))] += 1
    return (cells > 0).sum() / 10 > 0.7


def _extract_3d_line(pos, DU, rng):
    n = len(pos)
    max_iter = min(10, int(n * (n - 1) * 0.5))
    idx = list(range(n))
    best, bestA, bestB = [], None, None
    for _ in range(max_iter):
        left = n
        for k in range(2):
            r = rng() % left
            idx[k], idx[k + r] = idx[k + r], idx[k]
            left -= 1
        A, B = pos[idx[0]], pos[idx[1]]
        if np.linalg.norm(B - A) < 1e-10:
            continue
        inl = [i for i in range(n) if _mah_dist(pos[i], DU[i], A, B) < 1.5]
        if len(inl) > len(best) and _verify(pos[inl], A, B):
            best, bestA, bestB = inl, A, B
        if len(best) > n * 0.6:
            break
    A = B = np.zeros(3)
    if len(best) >= 2:
        m, d = (bestA + bestB) * 0.5, bestB - bestA
        while True:
            tm = pos[best].mean(0) if False else pos[best].sum(0) * (1.0 / len(best))
            td = np.linalg.svd(pos[best] - tm)[2][0]
            tmp = [i for i in range(n) if _mah_dist(pos[i], DU[i], tm, tm + td) < 1.5]
            if len(tmp) > len(best):
                best, m, d = tmp, tm, td
            else:
                break
        dp = (pos[best] - m) @ d
        e1, e2, minv, maxv = 0, 0, 100.0, -100.0
        for i, x in enumerate(dp):
            if x < minv:
                minv, e1 = x, i
            if x > maxv:
                maxv, e2 = x, i
        A, B = pos[best[e1]], pos[best[e2]]
    return A, B, len(best)


def is_line_good(lines, depth_f32, K32, k_as_f64, cx, cy, invfx, invfy, seed=1):
    """Returns (mvDepthLine float32[n], mvLines3D float64[n,6], inlier counts int32[n])."""
    f = float(np.float32(K32[0])) if k_as_f64 else focal_as_reference_reads_it(K32)
    rng = GlibcRand(seed)
    h, w = depth_f32.shape
    n = len(lines)
    depth_line = np.full(n, -1.0, np.float32)
    l3d = np.zeros((n, 6))
    ninl = np.zeros(n, np.int32)
    cx, cy, invfx, invfy = np.float32(cx), np.float32(cy), np.float32(invfx), np.float32(invfy)
    for i in range(n):
        sx, sy = np.float32(lines["start_point_x"][i]), np.float32(lines["start_point_y"][i])
        ex, ey = np.float32(lines["end_point_x"][i]), np.float32(lines["end_point_y"][i])
        dx, dy = np.float32(sx - ex), np.float32(sy - ey)
        length = float(np.sqrt(float(dx) * float(dx) + float(dy) * float(dy)))
        num = float(min(int(length), 50))
        if not num >= 1:
            continue
        pts = []
        for j in range(int(num) + 1):
            t = j / num
            px = np.float32(np.float32(float(sx) * (1 - t)) + np.float32(float(ex) * t))
            py = np.float32(np.float32(float(sy) * (1 - t)) + np.float32(float(ey) * t))
            x, y = float(px), float(py)
            if x < 0 or y < 0 or x >= w or y >= h:
                continue
            if np.floor(x) == x and np.floor(y) == y:
                col, row = max(int(x - 1), 0), max(int(y - 1), 0)
            else:
                col, row = int(x), int(y)
            d = depth_f32[row, col]
            if float(d) <= 0.01:
                continue
            z = float(d)
            pts.append([float(np.float32(np.float32(col) - cx)) * z * float(invfx),
                        float(np.float32(np.float32(row) - cy)) * z * float(invfy), z])
        if len(pts) < 10:
            continue
        pos = np.array(pts)
        DU = np.stack([_comp_pt3d_cov(p, f) for p in pos])
        A, B, k = _extract_3d_line(pos, DU, rng)
        ninl[i] = k
        if k / length > 0.4 and np.linalg.norm(A - B) > 0.02:
            depth_line[i] = min(depth_f32[int(ey), int(ex)], depth_f32[int(sy), int(sx)])
            l3d[i] = np.concatenate([A, B])
    return depth_line, l3d, ninl
