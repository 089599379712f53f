#!/bin/bash
# Round-6 evidence (run through gpurun from the repo root):  bash tools/profile_r06.sh
#  1. tools/profile_round.sh r06 512: kernel statistics + counter passes of the headline step (folded by tools/fold_profiles.sh r06 512)
#  2. kernel statistics of one 512-frame step of the lines and planes batch paths (alone) and their lane-utilisation counters
#  3. kernel statistics of the same entries at BASELINE config 5's size (1280 x 960, 128 frames per call)
#  4. the paths' saturation with steps in flight
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r06
bash $R/tools/profile_round.sh $TAG 512 > $OUT/${TAG}_profile_round.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $R
for path in lines planes; do
  T=$([ $path = lines ] && echo 2 || echo 6)
  rm -rf $OUT/${TAG}_stats_$path $OUT/${TAG}_pmc2_$path
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_$path -o $path -- python3 tools/path_saturation.py $path $T 1 > $OUT/${TAG}_stats_$path.log 2>&1
  find $OUT/${TAG}_stats_$path -name "*kernel_trace.csv" -delete
  timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc2_$path -o p -- python3 tools/path_saturation.py $path $T 1 > $OUT/${TAG}_pmc2_$path.log 2>&1
done
rm -rf $OUT/${TAG}_stats_c5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_c5 -o c5 -- python3 tools/config5_long_kernels.py 128 > $OUT/${TAG}_stats_c5.log 2>&1
find $OUT/${TAG}_stats_c5 -name "*kernel_trace.csv" -delete
{
  echo "== one path alone, steps in flight (tools/path_saturation.py) =="
  timeout -k 10 300 python3 tools/path_saturation.py lines 2 1 3 5 2>&1 | grep "lines:"
  timeout -k 10 300 python3 tools/path_saturation.py planes 6 1 3 5 2>&1 | grep "planes:"
  echo "== BASELINE config 5, 128 frames per call (tools/config5_long_kernels.py, no profiler) =="
  timeout -k 10 300 python3 tools/config5_long_kernels.py 128 2>&1 | tail -3
} > $OUT/${TAG}_path_saturation.txt 2>&1
for path in lines planes; do
  echo "== $path: kernel stats =="; head -10 $(find $OUT/${TAG}_stats_$path -name "*kernel_stats.csv" | head -1)
  echo "== $path: lane utilisation counters =="; python3 tools/pmc_summary.py $(find $OUT/${TAG}_pmc2_$path -name "*counter_collection.csv")
done > $OUT/${TAG}_long_kernels_summary.txt 2>&1
echo "== config 5: kernel stats ==" >> $OUT/${TAG}_long_kernels_summary.txt; head -12 $(find $OUT/${TAG}_stats_c5 -name "*kernel_stats.csv" | head -1) >> $OUT/${TAG}_long_kernels_summary.txt
cat $OUT/${TAG}_path_saturation.txt; cat $OUT/${TAG}_long_kernels_summary.txt
