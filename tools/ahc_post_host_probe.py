"""Host time of Frame::ComputePlanes' per-plane loop (drfe_planes_ahc_postprocess, ctx = NULL) on the AHC planes of one synthetic
frame, no device needed.  DRFE_TRACE_PLANES=1 prints gather / voxel grid / refit per call."""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.')
from dr_slam_amd import lib, synth
from oracle import oracle as O
cam = synth.ICL
_, d, _ = next(synth.sequence(3, 1, kind="living_room"))
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
o = O.ahc_planes(d, K4, inv)
g = lib.planes_ahc_from_blocks(o["blocks"], o["block_valid"], o["block_N"], d, K4, inv)
L = lib.load()
planes = np.ascontiguousarray(g["planes"], lib.PLANE_DTYPE); n = len(planes)
off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(m) for m in g["members"]])
idx = np.ascontiguousarray(np.concatenate([np.asarray(m, np.int32) for m in g["members"]]), np.int32)
post = np.zeros(n, lib.PLANE_POST_DTYPE); vox = np.zeros((len(idx), 3), np.float32); voff = np.zeros(n + 1, np.int32)
na, pn = C.c_int(), C.c_int()
p = lambda a: a.ctypes.data_as(C.c_void_p)
h, w = d.shape
def run():
    rc = L.drfe_planes_ahc_postprocess(None, p(d), w, h, w, p(K4), np.float32(inv), p(planes), n, p(off), p(idx), np.float32(9.0), 0.10,
                                       p(post), p(vox), p(voff), len(vox), C.byref(na), C.byref(pn))
    assert rc == 0
run()
t = time.perf_counter()
for _ in range(10): run()
print("post-processing %.2f ms per frame; %d planes, %d accepted, %d member points" % ((time.perf_counter() - t) / 10 * 1e3, n, na.value, len(idx)))
