"""One configuration of bench.py's full_frontend for a profiler: python tools/ff_once.py [frames] (DRFE_FF_INFLIGHT / DRFE_FF_SPLIT apply)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
r = bench.full_frontend("ICL", n_frames=n, reps=2)
print(json.dumps({k: r[k] for k in ("value", "frames_per_step", "steps_in_flight", "ms_per_step", "host_cpu_ms_per_frame", "host_cpu_utilisation", "host_cpu_ms_per_frame_by_pool", "host_threads_per_step_in_flight", "stage_wall_ms_last_step")}), flush=True)
