# full front-end (512 frames per step, 4 in flight unless DRFE_FF_INFLIGHT is set) under values of one environment variable:
#   bash tools/ff_env_sweep.sh VAR "v1 v2 ..." [repeats]
for r in $(seq 1 ${3:-1}); do for v in $2; do
  echo "$1=$v"; env $1=$v python tools/full_frontend_sweep.py 512 2>&1 | grep -v amdgpu | cut -c1-330
done; done
