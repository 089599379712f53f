import os, sys, time, numpy as np
sys.path.insert(0, '.')
from dr_slam_amd import lib, sharding, synth
cam = synth.ICL
base = sharding.render_sequence(3, 8, cam, "living_room", workers=1)
c = lib.Context(max_batch=1)
for i in range(3): c.lsd_extract(base[i][0])
os.environ["X"]="1"
t0=time.perf_counter()
for i in range(20): c.lsd_extract(base[i % 8][0])
print("lsd_extract: %.2f ms per frame" % ((time.perf_counter()-t0)*1e3/20))
