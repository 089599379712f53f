"""Seeded scenarios for LSDmatcher::SearchByProjection, shared by the CPU oracle tests and the
-m gpu parity tests.  Every map line is aimed at one current key line (midpoint, slope and descriptor perturbed), so
the GetLinesInArea gates, the best / second-best scan, the ratio test and the sequential claims all fire."""
import numpy as np

CAM = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, bf=40.0, min_x=0.0, max_x=640.0, min_y=0.0, max_y=480.0)
SCALE = (1.2 ** np.arange(8)).astype(np.float32)


def _pose(rng, tz):
    a = rng.uniform(-0.03, 0.03, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), tz]
    return T.astype(np.float32)


def _flip(rng, desc, k):
    out = desc.copy()
    for b in rng.choice(256, size=k, replace=False):
        out[b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def make(seed, keyline_dtype, mapline_dtype, tracked_dtype, n_cur=40, n_last=48, motion=0.0):
    """motion: z translation of the last camera relative to the current one (> mb -> forward, < -mb -> backward)."""
    rng = np.random.RandomState(seed)
    cur = np.zeros(n_cur, keyline_dtype)
    cur["pt_x"] = rng.uniform(60, 580, n_cur).astype(np.float32)
    cur["pt_y"] = rng.uniform(60, 420, n_cur).astype(np.float32)
    cur["angle"] = rng.uniform(-1.2, 1.2, n_cur).astype(np.float32)
    cur["octave"] = (rng.uniform(size=n_cur) < 0.25).astype(np.int32)
    cur_desc = rng.randint(0, 256, (n_cur, 32)).astype(np.uint8)
    Tcw_cur = _pose(rng, 0.0)
    Tcw_last = _pose(rng, motion)
    Rcw = Tcw_cur[:3, :3].astype(np.float64)
    tcw = Tcw_cur[:3, 3].astype(np.float64)
    last = np.zeros(n_last, mapline_dtype)
    tracked = np.zeros(n_last, tracked_dtype)
    for i in range(n_last):
        j = rng.randint(n_cur)
        mid = np.array([cur["pt_x"][j], cur["pt_y"][j]], np.float64) + rng.normal(0, 4.0, 2)
        m = float(cur["angle"][j]) + rng.uniform(-0.35, 0.2)
        d = np.array([1.0, m]) / np.hypot(1.0, m) * rng.uniform(20, 60)
        p1, p2 = mid - d, mid + d
        z1, z2 = rng.uniform(1, 4, 2)
        pts = []
        for (u, v), z in ((p1, z1), (p2, z2)):
            Xc = np.array([(u - CAM["cx"]) * z / CAM["fx"], (v - CAM["cy"]) * z / CAM["fy"], z])
            pts.append(Rcw.T @ (Xc - tcw))
        k = int(rng.choice([0, 3, 10, 25, 40, 70, 120]))
        desc = _flip(rng, cur_desc[j], k)
        last["valid"][i] = rng.uniform() < 0.9
        last["octave"][i] = int(cur["octave"][j]) if rng.uniform() < 0.8 else 1 - int(cur["octave"][j])
        last["obs_positive"][i] = rng.uniform() < 0.8
        last["world"][i] = np.concatenate(pts)
        last["desc"][i] = desc
        tracked["in_view"][i] = rng.uniform() < 0.9
        tracked["level"][i] = last["octave"][i]
        tracked["obs_positive"][i] = last["obs_positive"][i]
        tracked["x1"][i], tracked["y1"][i] = np.float32(p1[0]), np.float32(p1[1])
        tracked["x2"][i], tracked["y2"][i] = np.float32(p2[0]), np.float32(p2[1])
        tracked["view_cos"][i] = np.float32(rng.uniform(0.99, 1.0))
        tracked["desc"][i] = desc
    cur_ml = np.full(n_cur, -1, np.int32)
    cur_obs = np.zeros(n_cur, np.uint8)
    for j in rng.choice(n_cur, size=max(1, n_cur // 8), replace=False):
        cur_ml[j] = 1000 + j                        # pre-existing claims, half of them overridable
        cur_obs[j] = rng.uniform() < 0.5
    return dict(cur=cur, cur_desc=cur_desc, Tcw_cur=Tcw_cur, Tcw_last=Tcw_last, last=last, tracked=tracked,
                cur_ml=cur_ml, cur_obs=cur_obs)


def cam9():
    return np.array([CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], np.float32(CAM["bf"]) / np.float32(CAM["fx"]),
                     CAM["min_x"], CAM["max_x"], CAM["min_y"], CAM["max_y"]], np.float32)


def sim3_line_scene(seed, n_kf, n, keyline_dtype, mapline_dtype, tracked_dtype, frustum_line_dtype):
    """Map lines seen from a keyframe, with normals / distance bands that exercise every gate (as test_lsd_fuse_search)."""
    sc = make(seed, keyline_dtype, mapline_dtype, tracked_dtype, n_cur=n_kf, n_last=n)
    rng = np.random.RandomState(300 + seed)
    Tcw = sc["Tcw_cur"]
    Twc = np.linalg.inv(Tcw.astype(np.float64))
    lines = np.zeros(n, frustum_line_dtype)
    lines["world"] = sc["last"]["world"]
    mid = 0.5 * (lines["world"][:, :3] + lines["world"][:, 3:])
    om = mid - Twc[:3, 3][None, :]
    dist = np.linalg.norm(om, axis=1)
    nrm = om / dist[:, None] + rng.normal(0, 0.5, (n, 3))
    lines["normal"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    lvl = rng.choice([-1, 0, 1, 2, 9], size=n, p=[0.04, 0.4, 0.4, 0.12, 0.04])
    lines["max_distance"] = (dist * 1.2 ** (lvl - rng.uniform(0.1, 0.9, n))).astype(np.float32)
    lines["min_distance"] = (dist * rng.uniform(0.3, 1.3, n)).astype(np.float32)
    return sc, lines, rng



def keyframe_with_own_map_lines(seed, n, keyline_dtype, frustum_line_dtype):
    """A keyframe whose key line i carries map line i (aimed at it: midpoint, slope and descriptor perturbed), as
    KeyFrame::GetMapLineMatches() presents them."""
    rng = np.random.RandomState(700 + seed)
    kl = np.zeros(n, keyline_dtype)
    kl["pt_x"] = rng.uniform(80, 560, n).astype(np.float32)
    kl["pt_y"] = rng.uniform(80, 400, n).astype(np.float32)
    kl["angle"] = rng.uniform(-1.2, 1.2, n).astype(np.float32)
    kl["octave"] = (rng.uniform(size=n) < 0.25).astype(np.int32)
    kdesc = rng.randint(0, 256, (n, 32)).astype(np.uint8)
    Tcw = _pose(rng, 0.0)
    Rcw, tcw = Tcw[:3, :3].astype(np.float64), Tcw[:3, 3].astype(np.float64)
    lines = np.zeros(n, frustum_line_dtype)
    descs = np.zeros((n, 32), np.uint8)
    for i in range(n):
        mid = np.array([kl["pt_x"][i], kl["pt_y"][i]], np.float64) + rng.normal(0, 3.0, 2)
        m = float(kl["angle"][i]) + rng.uniform(-0.3, 0.1)
        d = np.array([1.0, m]) / np.hypot(1.0, m) * rng.uniform(20, 50)
        pts = []
        for (u, v), z in ((mid - d, rng.uniform(1, 4)), (mid + d, rng.uniform(1, 4))):
            Xc = np.array([(u - CAM["cx"]) * z / CAM["fx"], (v - CAM["cy"]) * z / CAM["fy"], z])
            pts.append(Rcw.T @ (Xc - tcw))
        lines["world"][i] = np.concatenate(pts)
        descs[i] = _flip(rng, kdesc[i], int(rng.choice([0, 3, 10, 25, 40, 70, 120])))
    mid = 0.5 * (lines["world"][:, :3] + lines["world"][:, 3:])
    dist = np.linalg.norm(mid - np.linalg.inv(Tcw.astype(np.float64))[:3, 3][None, :], axis=1)
    lvl = np.where(rng.uniform(size=n) < 0.05, 9, kl["octave"] + (rng.uniform(size=n) < 0.3))
    lines["max_distance"] = (dist * 1.2 ** (lvl - rng.uniform(0.1, 0.9, n))).astype(np.float32)
    lines["min_distance"] = (dist * rng.uniform(0.3, 1.1, n)).astype(np.float32)
    lines["normal"] = [0, 0, 1]
    return Tcw, kl, kdesc, lines, descs, rng



def keyframe_pair_for_sim3(seed, keyline_dtype, frustum_line_dtype):
    """Two keyframes that see the same n world lines from nearby poses (key line i of either carries its own map line; KF2 lists
    them in another order) and the similarity between them: everything LSDmatcher::SearchBySim3 reads."""
    n = 40 if seed < 10 else 300
    T1w, kl1, kd1, lines1, descs1, rng = keyframe_with_own_map_lines(seed, n, keyline_dtype, frustum_line_dtype)
    # keyframe 2: a nearby pose looking at the same world lines; its key lines are where they project (perturbed), in another order
    ang = np.deg2rad(rng.uniform(-2, 2, 2))
    Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
    Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
    T21 = np.eye(4); T21[:3, :3] = Rx @ Ry; T21[:3, 3] = rng.uniform(-0.05, 0.05, 3)
    T2w = (T21 @ T1w.astype(np.float64)).astype(np.float32)
    perm = rng.permutation(n)                                    # key line k of KF2 shows world line perm[k]
    W = lines1["world"][perm]
    kl2 = np.zeros(n, keyline_dtype)
    R2, t2 = T2w[:3, :3].astype(np.float64), T2w[:3, 3].astype(np.float64)
    a = (R2 @ W[:, :3].T).T + t2; b = (R2 @ W[:, 3:].T).T + t2
    ua = CAM["fx"] * a[:, 0] / a[:, 2] + CAM["cx"]; va = CAM["fy"] * a[:, 1] / a[:, 2] + CAM["cy"]
    ub = CAM["fx"] * b[:, 0] / b[:, 2] + CAM["cx"]; vb = CAM["fy"] * b[:, 1] / b[:, 2] + CAM["cy"]
    kl2["pt_x"] = (0.5 * (ua + ub) + rng.normal(0, 2.0, n)).astype(np.float32)
    kl2["pt_y"] = (0.5 * (va + vb) + rng.normal(0, 2.0, n)).astype(np.float32)
    kl2["angle"] = ((va - vb) / (ua - ub) + rng.uniform(0.0, 0.3, n)).astype(np.float32)
    kl2["octave"] = kl1["octave"][perm]
    kd2 = np.stack([_flip(rng, kd1[perm[k]], int(rng.choice([0, 5, 20, 60]))) for k in range(n)])
    lines2 = lines1[perm].copy(); descs2 = np.stack([_flip(rng, descs1[perm[k]], int(rng.choice([0, 4, 12]))) for k in range(n)])
    T12 = np.linalg.inv(T21)
    s12 = float(1.0 + rng.uniform(-0.03, 0.03))
    R12 = T12[:3, :3].astype(np.float32); t12 = T12[:3, 3].astype(np.float32)
    skip1 = (rng.uniform(size=n) < 0.15).astype(np.uint8); skip2 = (rng.uniform(size=n) < 0.15).astype(np.uint8)
    return dict(n=n, T1w=T1w, T2w=T2w, s12=s12, R12=R12, t12=t12, perm=perm, lines1=lines1, descs1=descs1, skip1=skip1, kl1=kl1, kd1=kd1,
                lines2=lines2, descs2=descs2, skip2=skip2, kl2=kl2, kd2=kd2)


# ------------------------------------------------------------------------------------------------
# Analytic line scene: convex polygons of known corners, area-sampled.  Every polygon edge is one step edge of known
# position, direction and length, so what LSD must report is known in closed form (independent of any LSD code).

def _coverage(h, w, pts, ss=4):
    """fraction of each pixel covered by the convex polygon `pts` (pixel-centre coordinates), ss x ss supersampling"""
    ys = (np.arange(h * ss) + 0.5) / ss - 0.5
    xs = (np.arange(w * ss) + 0.5) / ss - 0.5
    X, Y = np.meshgrid(xs, ys)
    P = np.asarray(pts, float)
    c = P.mean(0)
    inside = np.ones_like(X, bool)
    for a, b in zip(P, np.roll(P, -1, 0)):
        nrm = np.array([-(b - a)[1], (b - a)[0]])
        s = np.sign(nrm @ (c - a))
        inside &= s * ((X - a[0]) * nrm[0] + (Y - a[1]) * nrm[1]) >= 0
    return inside.reshape(h, ss, w, ss).mean((1, 3))


def _rot_rect(cx, cy, a, b, deg):
    t = np.deg2rad(deg)
    R = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
    return (np.array([[-a, -b], [a, -b], [a, b], [-a, b]], float) @ R.T) + [cx, cy]


def analytic_polygons(h=480, w=640):
    """-> (gray uint8 image, list of edges (a, b) in image coordinates): two rotated rectangles and a triangle on a flat
    background, 11 edges between 138 and 270 px long"""
    polys = [_rot_rect(180, 150, 110, 70, 17.0), _rot_rect(450, 300, 120, 90, -31.0),
             np.array([[330.0, 40.0], [600.0, 70.0], [420.0, 170.0]])]
    img = np.full((h, w), 60.0)
    for p, v in zip(polys, (200.0, 150.0, 230.0)):
        m = _coverage(h, w, p)
        img = img * (1 - m) + v * m
    edges = [(a, b) for p in polys for a, b in zip(p, np.roll(p, -1, 0))]
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), edges
