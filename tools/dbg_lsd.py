import sys; sys.path.insert(0, '/root/repo')
from dr_slam_amd import lib, synth
g,_,_ = next(synth.sequence(5,1,kind="corridor"))
c = lib.Context(max_batch=1)
a = c.lsd_extract(g)
print(a["detected"])
