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
