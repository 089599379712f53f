"""Phase times inside k_lsd_order (a -DORD_PROFILE build: make -C dr_slam_amd/csrc variant NAME=ordprof DEF=-DORD_PROFILE
SRC=lsd_order_kernels.hip; run with DRFE_LIB=build/libdrfe_ordprof.so): 512 living-room frames through drfe_lsd_extract_batch."""
import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(512, len(base))
gray = np.stack([base[i][0] for i in order])
ctx = lib.Context(max_batch=1)
ctx.lsd_extract_batch(gray, n_threads=2)
L = ctypes.CDLL(os.environ["DRFE_LIB"])
out = (ctypes.c_ulonglong * 9)()
L.drfe_debug_order_profile(out)
ctx.lsd_extract_batch(gray, n_threads=2)
L.drfe_debug_order_profile(out)
v = list(out)
print("k_lsd_order, per frame (ms of the workgroup): workgroup partitions %.2f, wavefront phase %.2f, counting passes %.2f; %d frames" % (v[0] / 1e5 / v[3], v[1] / 1e5 / v[3], v[2] / 1e5 / v[3], v[3]))
print("  wavefront 0 of each frame (ms): partitions in HBM %.2f, copy into LDS %.2f, wavefront partitions in LDS %.2f, one range per lane %.2f, copy back %.2f" % tuple(x / 1e5 / v[3] for x in v[4:9]))
ctx.close()
