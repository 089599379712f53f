import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
r = bench.full_frontend("ICL", n_frames=512, reps=8)
print(os.environ.get("DRFE_LSD_GROW_WAVES", "4"), round(r["value"]), r["ms_per_step"], flush=True)
