"""Host time of the sequential half of LSD (drfe_lsd_segments_host: pixel ordering, region growing, rectangle fit, NFA with
host pixel counts) on the oracle's level-line fields of one synthetic frame, no device needed.  DRFE_TRACE_LINES=1 prints the
stage times."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from dr_slam_amd import lib, synth
from oracle import oracle as O
g, _, _ = next(synth.sequence(3, 1, cam=synth.ICL, kind="living_room"))
o = O.extract_lines(g, max_lines=100000, stages=True)
ang = o["angles"]
cs = np.zeros(ang.shape + (2,), np.float32)
defined = ang != -1024.0
a32 = ang.astype(np.float32)
cs[..., 0] = np.where(defined, np.cos(a32.astype(np.float64)).astype(np.float32), 0)
cs[..., 1] = np.where(defined, np.sin(a32.astype(np.float64)).astype(np.float32), 0)
mx = float(o["modgrad"].max())
segs = lib.lsd_segments_host(o["modgrad"], ang, cs, mx)
t = time.perf_counter()
for _ in range(10): segs = lib.lsd_segments_host(o["modgrad"], ang, cs, mx)
print("LSD host %.2f ms per frame; %d segments (oracle %d)" % ((time.perf_counter() - t) / 10 * 1e3, len(segs), o["detected"]))
