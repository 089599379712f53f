"""Seeded synthetic RGB-D sequences: the reference ships no dataset, so every config
is synthetic.  A scene is an axis-aligned room with boxes, textured procedurally *in surface
coordinates* (so consecutive frames are matchable), ray-cast through a pinhole camera.

gray  : uint8 (h, w)     the primary input of the path (Tracking::GrabImageRGBD converts upstream,
                          reference src/Tracking.cc:197-207)
depth : uint16 (h, w)    round(z * DepthMapFactor) with +-LSB noise, ~2 % zero holes, a wall beyond 5 m
                          (exercises the z>5.0 clamp, reference src/PlaneExtractor.cpp:44)
Everything is integer hashing (SplitMix64) + float64 ray casting; no global RNG state.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised SplitMix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _hash3(seed: int, a: np.ndarray, b: np.ndarray, c) -> np.ndarray:
    with np.errstate(over="ignore"):
        k = (a.astype(np.int64).astype(np.uint64) * np.uint64(0x100000001B3)
             + b.astype(np.int64).astype(np.uint64) * np.uint64(0x9E3779B1)
             + np.uint64(c) * np.uint64(0x85EBCA77)
             + np.uint64(seed) * np.uint64(0xC2B2AE3D27D4EB4F))
    return splitmix64(k)


@dataclass
class Camera:
    fx: float
    fy: float
    cx: float
    cy: float
    bf: float = 40.0
    depth_factor: float = 5000.0  # DepthMapFactor (raw units per metre)
    w: int = 640
    h: int = 480
    dist: tuple = ()              # Camera.k1, k2, p1, p2, k3 (mDistCoef); empty / k1 == 0: no undistortion

    def scaled(self, s: float) -> "Camera":
        return Camera(self.fx * s, self.fy * s, self.cx * s, self.cy * s, self.bf * s, self.depth_factor,
                      int(round(self.w * s)), int(round(self.h * s)), self.dist)


# intrinsics of reference Examples/RGB-D/*.yaml (TUM3.yaml:8-34, ICL.yaml:8-11, Realsense.yaml:8-17)
TUM3 = Camera(535.4, 539.2, 320.1, 247.6, 40.0, 5000.0)
# the two settings with lens distortion (TUM1.yaml:8-35, TUM2.yaml:8-35): Frame::UndistortKeyPoints is live
TUM1 = Camera(517.306408, 516.469215, 318.643040, 255.313989, 40.0, 5000.0, 640, 480,
              (0.262383, -0.953104, -0.005358, 0.002628, 1.163314))
TUM2 = Camera(520.908620, 521.007327, 325.141442, 249.701764, 40.0, 5208.0, 640, 480,
              (0.231222, -0.784899, -0.003257, -0.000105, 0.917205))
ICL = Camera(481.2, -480.0, 319.5, 239.5, 40.0, 5000.0)
REALSENSE = Camera(615.9, 616.1, 323.0, 241.5, 30.8, 1000.0)


@dataclass
class Scene:
    seed: int
    kind: str = "room_boxes"  # room_boxes | planar_lowtexture | living_room | corridor
    boxes: list = field(default_factory=list)

    def __post_init__(self):
        h = splitmix64(np.arange(64, dtype=np.uint64) + np.uint64(self.seed * 1000003))
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        nb = {"room_boxes": 5, "living_room": 7, "corridor": 3, "planar_lowtexture": 0}[self.kind]
        self.half = (3.0, 1.4, 6.2) if self.kind != "corridor" else (1.2, 1.4, 8.0)
        self.boxes = []
        for i in range(nb):
            cx = (u[4 * i] - 0.5) * 2.0 * (self.half[0] - 0.7)
            cz = 1.5 + u[4 * i + 1] * 2.6
            sx = 0.25 + 0.45 * u[4 * i + 2]
            sy = 0.3 + 0.9 * u[4 * i + 3]
            self.boxes.append(((cx - sx, self.half[1] - sy, cz - sx), (cx + sx, self.half[1], cz + sx)))


def pose_twc(k: int, step_m: float = 0.01, yaw_deg: float = 0.2) -> np.ndarray:
    """Camera-to-world pose of frame k: translate along +x, yaw about the (downward) y axis."""
    a = math.radians(yaw_deg * k)
    T = np.eye(4)
    T[:3, :3] = [[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]
    T[:3, 3] = [step_m * k - 0.3, 0.0, -1.0]
    return T


def _texture(scene: Scene, face: np.ndarray, s: np.ndarray, t: np.ndarray) -> np.ndarray:
    low = scene.kind == "planar_lowtexture"
    out = np.full(s.shape, 100.0)
    scales = ((0.5, 50), (0.12, 60), (0.035, 50)) if not low else ((0.9, 34), (0.3, 14))
    for lvl, (cell, amp) in enumerate(scales):
        a = np.floor(s / cell).astype(np.int64)
        b = np.floor(t / cell).astype(np.int64)
        hv = _hash3(scene.seed, a * 64 + face.astype(np.int64), b, lvl + 1)
        out += ((hv >> np.uint64(20)) % np.uint64(1024)).astype(np.float64) / 1023.0 * amp - amp / 2
    return out


def render(scene: Scene, cam: Camera, Twc: np.ndarray, frame_id: int = 0):
    """Ray-cast one frame -> (gray uint8 [h,w], depth uint16 [h,w])."""
    h, w = cam.h, cam.w
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    d_cam = np.stack([(u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u)], -1)
    R, o = Twc[:3, :3], Twc[:3, 3]
    d = d_cam @ R.T
    best_t = np.full((h, w), np.inf)
    best_face = np.zeros((h, w), np.int64)

    def slab(lo, hi, inside, face_base):
        nonlocal best_t, best_face
        with np.errstate(divide="ignore", invalid="ignore"):
            for ax in range(3):
                for side, plane in ((0, lo[ax]), (1, hi[ax])):
                    tt = (plane - o[ax]) / d[..., ax]
                    p = o + d * tt[..., None]
                    ok = tt > 1e-6
                    for a2 in range(3):
                        if a2 != ax:
                            ok &= (p[..., a2] >= lo[a2] - 1e-9) & (p[..., a2] <= hi[a2] + 1e-9)
                    # room: hit walls from inside; boxes: hit faces from outside
                    facing = (d[..., ax] > 0) if (side == 1) == inside else (d[..., ax] < 0)
                    ok &= facing & (tt < best_t)
                    best_t = np.where(ok, tt, best_t)
                    best_face = np.where(ok, face_base + ax * 2 + side, best_face)

    hx, hy, hz = scene.half
    slab((-hx, -hy, -2.5), (hx, hy, hz), True, 0)
    for bi, (lo, hi) in enumerate(scene.boxes):
        slab(lo, hi, False, 8 * (bi + 1))
    hit = np.isfinite(best_t)
    tt = np.where(hit, best_t, 1.0)
    p = o + d * tt[..., None]
    ax = (best_face % 8) // 2
    s = np.where(ax == 0, p[..., 1], p[..., 0])
    t = np.where(ax == 2, p[..., 1], p[..., 2])
    g = _texture(scene, best_face, s, t)
    g *= np.clip(1.15 - 0.05 * tt, 0.6, 1.2)  # mild distance shading
    xi = np.arange(w, dtype=np.int64)[None, :].repeat(h, 0)
    yi = np.arange(h, dtype=np.int64)[:, None].repeat(w, 1)
    nz = _hash3(scene.seed ^ 0x5555, xi, yi, frame_id + 17)
    g += (nz % np.uint64(7)).astype(np.float64) - 3.0
    gray = np.clip(np.rint(g), 0, 255).astype(np.uint8)
    gray[~hit] = 0
    z = tt * d_cam[..., 2]  # depth along the optical axis (d_cam z == 1)
    dz = (nz >> np.uint64(8)) % np.uint64(5)
    raw = np.rint(z * cam.depth_factor) + dz.astype(np.float64) - 2.0
    raw = np.clip(raw, 0, 65535)
    # ~2 % missing depth, in 6x6-pixel patches (sensor dropouts are clustered; i.i.d. pixel holes would
    # invalidate ~87 % of PEAC's 10x10 init blocks under INIT_STRICT)
    hole = (_hash3(scene.seed ^ 0x7777, xi // 6, yi // 6, frame_id + 31) % np.uint64(50)) == 0
    raw[hole | ~hit] = 0
    return gray, raw.astype(np.uint16)


def sequence(seed: int, n_frames: int, cam: Camera = TUM3, kind: str = "room_boxes", start: int = 0):
    """Yield (gray, depth, Twc) for frames start..start+n_frames-1 of a deterministic sequence."""
    sc = Scene(seed, kind)
    for k in range(start, start + n_frames):
        T = pose_twc(k)
        g, dpt = render(sc, cam, T, k)
        yield g, dpt, T


def noise_frame(seed: int, w: int, h: int) -> np.ndarray:
    """Dense random rectangles + noise (no 3-D consistency): worst-case corner density for stress tests."""
    xi = np.arange(w, dtype=np.int64)[None, :].repeat(h, 0)
    yi = np.arange(h, dtype=np.int64)[:, None].repeat(w, 1)
    g = np.zeros((h, w))
    for lvl, cell in enumerate((37, 11, 5)):
        hv = _hash3(seed, xi // cell, yi // cell, lvl + 3)
        g += (hv % np.uint64(90)).astype(np.float64)
    g += (_hash3(seed, xi, yi, 99) % np.uint64(9)).astype(np.float64)
    return np.clip(g, 0, 255).astype(np.uint8)
