#!/bin/bash
# The evidence for the long (one-wavefront-per-frame and sort) kernels of the batch paths, collected on the GPU box through gpurun
# from the repo root:  bash tools/profile_long_kernels.sh <tag>
# Needs the profile variants (make -C dr_slam_amd/csrc variant NAME=lsdprof DEF=-DLSD_PROFILE SRC=lsd_grow_kernels.hip, ahcprof /
# -DAHC_PROFILE / ahc_frame_kernels.hip, ordprof / -DORD_PROFILE / lsd_order_kernels.hip, voxprof / -DVOX_PROFILE / voxel_kernels.hip).
# Writes gpurun_out/<tag>_*: phase timers, kernel statistics of one 512-frame step of each path, SQ counters, path saturation.
set -u
TAG=${1:-r04b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $R
{
  echo "== phase timers of the profile builds (frame 0 of a 512-frame step; the kernel alone on the device) =="
  DRFE_LIB=$R/build/libdrfe_lsdprof.so DRFE_LSD_PROFILE=1 timeout -k 10 200 python3 tools/path_saturation.py lines 2 1 2>&1 | grep -E "k_lsd_grow frame" | tail -3
  DRFE_LIB=$R/build/libdrfe_ordprof.so timeout -k 10 200 python3 tools/order_profile.py 2>&1 | tail -2
  DRFE_LIB=$R/build/libdrfe_ahcprof.so DRFE_AHC_PROFILE=1 timeout -k 10 200 python3 tools/path_saturation.py planes 6 1 2>&1 | grep -iE "k_ahc|ahCluster|flood fill" | tail -3
  DRFE_LIB=$R/build/libdrfe_voxprof.so timeout -k 10 200 python3 tools/voxel_profile.py 2>&1 | tail -11
} > $OUT/${TAG}_device_sequential_cores.txt 2>&1
{
  echo "== one path alone, steps in flight (tools/path_saturation.py) =="
  timeout -k 10 300 python3 tools/path_saturation.py lines 2 1 3 4 5 2>&1 | grep "lines:"
  timeout -k 10 300 python3 tools/path_saturation.py planes 6 1 3 4 5 2>&1 | grep "planes:"
  timeout -k 10 300 python3 tools/path_saturation.py cape 2 1 3 2>&1 | grep "cape:"
  echo "== one call of N line frames (tools/lines_big_batch.py) =="
  timeout -k 10 300 python3 tools/lines_big_batch.py 512 1024 2048 3072 2>&1 | grep "lines,"
  echo "== full front-end, steps in flight (tools/full_frontend_sweep.py) =="
  for n in 3 4 5 6; do DRFE_FF_INFLIGHT=$n timeout -k 10 300 python3 tools/full_frontend_sweep.py 512 2>&1 | grep value | cut -c1-400; done
} > $OUT/${TAG}_path_saturation.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_lines -o lines -- python3 tools/path_saturation.py lines 2 1 > $OUT/${TAG}_stats_lines.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_planes -o planes -- python3 tools/path_saturation.py planes 6 1 > $OUT/${TAG}_stats_planes.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_pmc_lines -o p -- python3 tools/path_saturation.py lines 2 1 > $OUT/${TAG}_pmc_lines.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_pmc_planes -o p -- python3 tools/path_saturation.py planes 6 1 > $OUT/${TAG}_pmc_planes.log 2>&1
ls $OUT | grep ${TAG}_
