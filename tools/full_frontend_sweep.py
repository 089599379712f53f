"""bench.py's full_frontend (BASELINE config 3) at several step sizes: python tools/full_frontend_sweep.py 96 256 512"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
for n in [int(a) for a in sys.argv[1:]] or [96, 256, 512]:
    r = bench.full_frontend("ICL", n_frames=n, reps=2)
    print(json.dumps({k: r[k] for k in ("value", "frames_per_step", "steps_in_flight", "ms_per_step", "host_cpu_ms_per_frame", "host_cpu_utilisation", "host_cpu_ms_per_frame_by_pool", "host_threads_per_step_in_flight", "stage_wall_ms_last_step")}), flush=True)
