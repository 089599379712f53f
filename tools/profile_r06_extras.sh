#!/bin/bash
# Round-6 side evidence (gpurun, repo root): phase profiles of the AHC kernels from the profile builds, the permlane micro-benchmark, the tail timeline.
# Needs: make -C dr_slam_amd/csrc variant NAME=ahcprof DEF=-DAHC_PROFILE SRC=ahc_frame_kernels.hip   (ahcchunks: "-DAHC_PROFILE -DAHC_PROFILE_CHUNKS",
#        ahchot: "-DAHC_PROFILE -DAHC_PROFILE_HOT", ahcff: -DAHC_PROFILE_FF) and tools/bin/ubench_permlane
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
{
  echo "== -DAHC_PROFILE, 640 x 480 (tools/path_saturation.py planes 6 1): phases of frame 0 =="
  DRFE_LIB=$R/build/libdrfe_ahcprof.so DRFE_AHC_PROFILE=1 timeout -k 10 200 python3 tools/path_saturation.py planes 6 1 2>&1 | grep -iE "k_ahc|ahCluster|flood fill" | tail -3
  echo "== -DAHC_PROFILE, 1280 x 960 (tools/config5_long_kernels.py 32) =="
  DRFE_LIB=$R/build/libdrfe_ahcprof.so DRFE_AHC_PROFILE=1 timeout -k 10 300 python3 tools/config5_long_kernels.py 32 2>&1 | grep -iE "k_ahc|ahCluster|flood fill" | tail -3
  echo "== -DAHC_PROFILE_CHUNKS: the last number is (neighbours summed over the pops) << 32 | trial chunks summed; 640 x 480, then 1280 x 960 =="
  DRFE_LIB=$R/build/libdrfe_ahcchunks.so DRFE_AHC_PROFILE=1 timeout -k 10 200 python3 tools/path_saturation.py planes 6 1 2>&1 | grep -iE "ahCluster" | tail -1
  DRFE_LIB=$R/build/libdrfe_ahcchunks.so DRFE_AHC_PROFILE=1 timeout -k 10 300 python3 tools/config5_long_kernels.py 32 2>&1 | grep -iE "ahCluster" | tail -1
  echo "== -DAHC_PROFILE_HOT: the last number counts the pops of the node the previous merge made (640 x 480) =="
  DRFE_LIB=$R/build/libdrfe_ahchot.so DRFE_AHC_PROFILE=1 timeout -k 10 200 python3 tools/path_saturation.py planes 6 1 2>&1 | grep -iE "ahCluster" | tail -1
  echo "== -DAHC_PROFILE_FF: shader cycles of frame 0's flood fill by phase (640 x 480; 3 150 steps of 32 entries) =="
  DRFE_LIB=$R/build/libdrfe_ahcff.so timeout -k 10 200 python3 tools/path_saturation.py planes 6 1 2>&1 | grep -iE "flood fill" | tail -1
} > $OUT/r06_ahc_phases.txt 2>&1
timeout -k 5 60 tools/bin/ubench_permlane > $OUT/r06_ubench_permlane.txt 2>&1
{ python3 tools/tail_timeline.py 20 3 2>&1 | tail -3; python3 tools/tail_timeline.py 21 3 2>&1 | tail -2; python3 tools/tail_timeline.py 20 5 2>&1 | tail -2; } > $OUT/r06_tail_timeline.txt 2>&1
cat $OUT/r06_ahc_phases.txt $OUT/r06_ubench_permlane.txt $OUT/r06_tail_timeline.txt
