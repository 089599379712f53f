"""Run any tool of this directory against a variant build of libdrfe.so (experiments only):
    python tools/run_variant.py <lib.so> <script.py> [args]"""
import os
import runpy
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dr_slam_amd.lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
