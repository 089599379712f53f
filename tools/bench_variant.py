"""Run bench.py against a variant build of libdrfe.so (experiments only): python tools/bench_variant.py <lib.so> [bench args]."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dr_slam_amd.lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
