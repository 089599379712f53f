"""Phase times inside k_voxel_grid (a -DVOX_PROFILE build: make -C dr_slam_amd/csrc variant NAME=voxprof DEF=-DVOX_PROFILE
SRC=voxel_kernels.hip; run with DRFE_LIB=build/libdrfe_voxprof.so): 512 living-room frames through drfe_planes_ahc_post_batch."""
import ctypes, os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
order = sharding.pingpong_order(512, len(base))
depth = np.stack([base[i][1] for i in order])
K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32); inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
ctx = lib.Context(max_batch=1)
ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=8)
out = (ctypes.c_ulonglong * 34)()
L = ctx.L if hasattr(ctx, "L") else lib._load()
L.drfe_debug_voxel_profile(out)
t = time.perf_counter()
ctx.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=8)
el = time.perf_counter() - t
L.drfe_debug_voxel_profile(out)
v = list(out)
names = ["bounds + keys", "workgroup partitions", "wavefront phase", "counting passes", "leaf heads", "centroids"]
tot = sum(v[:6])
print("512 frames in %.1f ms; %d plane clouds, %d points (%.0f per cloud); per cloud %.3f ms of workgroup time" % (el * 1e3, v[6], v[7], v[7] / max(1, v[6]), tot / 1e5 / max(1, v[6])))
for k, nme in enumerate(names):
    print("  %-22s %6.3f ms per cloud  %5.1f %%" % (nme, v[k] / 1e5 / max(1, v[6]), 100.0 * v[k] / max(1, tot)))
print("  longest workgroup %.2f ms; %d workgroups above 5 ms, %d above 20 ms; %d ended in the heap-sort flag after %.2f ms each" % (v[8] / 1e5, v[9], v[10], v[12], v[11] / 1e5 / max(1, v[12])))
print("  workgroup time per XCD (ms):", [round(x / 1e5) for x in v[16:24]], "non-empty workgroups per XCD:", v[24:32])
print("  non-empty workgroups running when one begins: %.1f on average" % (v[32] / max(1, v[6])))
ctx.close()
