#!/bin/bash
# Round-5 evidence for the long kernels of the batch paths (run through gpurun from the repo root):  bash tools/profile_r05.sh <tag>
# kernel statistics of one 512-frame step of the lines and planes paths (alone, and under load: five steps in flight / the full front-end), SQ counters (issue / wait) and the VALU lane-utilisation
# pair (SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU) in passes of their own, and the paths' saturation with steps in flight.
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $R
for path in lines planes; do
  T=$([ $path = lines ] && echo 8 || echo 6)
  rm -rf $OUT/${TAG}_stats_$path $OUT/${TAG}_pmc1_$path $OUT/${TAG}_pmc2_$path
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_$path -o $path -- python3 tools/path_saturation.py $path $T 1 > $OUT/${TAG}_stats_$path.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_pmc1_$path -o p -- python3 tools/path_saturation.py $path $T 1 > $OUT/${TAG}_pmc1_$path.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc2_$path -o p -- python3 tools/path_saturation.py $path $T 1 > $OUT/${TAG}_pmc2_$path.log 2>&1
done
{
  echo "== one path alone, steps in flight (tools/path_saturation.py) =="
  timeout -k 10 300 python3 tools/path_saturation.py lines 8 1 3 5 2>&1 | grep "lines:"
  timeout -k 10 300 python3 tools/path_saturation.py planes 6 1 3 5 2>&1 | grep "planes:"
  echo "== the same lines path with four wavefronts per frame forced at this batch size (DRFE_LSD_GROW_WAVES=4: k_lsd_grow_mw) =="
  DRFE_LSD_GROW_WAVES=4 timeout -k 10 300 python3 tools/path_saturation.py lines 8 1 3 5 2>&1 | grep "lines:"
} > $OUT/${TAG}_path_saturation.txt 2>&1
# the lines path with the four-wavefront growth kernel forced at this batch size (by default only calls of <= 256 frames take it)
rm -rf $OUT/${TAG}_stats_lines_mw
DRFE_LSD_GROW_WAVES=4 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_lines_mw -o lines -- python3 tools/path_saturation.py lines 8 1 > $OUT/${TAG}_stats_lines_mw.log 2>&1
# the long kernels UNDER LOAD: the lines path with five steps in flight, and the full front-end (bench.py's full_frontend, 2 x 5 steps)
rm -rf $OUT/${TAG}_load_lines $OUT/${TAG}_load_ff
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_load_lines -o l -- python3 tools/path_saturation.py lines 8 5 > $OUT/${TAG}_load_lines.log 2>&1
find $OUT/${TAG}_load_lines -name "*kernel_trace.csv" -delete
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_load_ff -o f -- python3 tools/ff_once.py > $OUT/${TAG}_load_ff.log 2>&1
find $OUT/${TAG}_load_ff -name "*kernel_trace.csv" -delete
for path in lines planes; do
  echo "== $path: kernel stats =="; head -8 $(find $OUT/${TAG}_stats_$path -name "*kernel_stats.csv" | head -1)
  echo "== $path: SQ counters =="; python3 tools/pmc_summary.py $(find $OUT/${TAG}_pmc1_$path -name "*counter_collection.csv")
  echo "== $path: lane utilisation counters =="; python3 tools/pmc_summary.py $(find $OUT/${TAG}_pmc2_$path -name "*counter_collection.csv")
done > $OUT/${TAG}_long_kernels_summary.txt 2>&1
cat $OUT/${TAG}_path_saturation.txt; cat $OUT/${TAG}_long_kernels_summary.txt
