# full front-end against the number of hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES, default 4) and steps in flight
for q in $1; do for n in $2; do
  echo "GPU_MAX_HW_QUEUES $q inflight $n"; GPU_MAX_HW_QUEUES=$q DRFE_FF_INFLIGHT=$n python tools/full_frontend_sweep.py 512 2>&1 | grep -v amdgpu | cut -c1-420
done; done
