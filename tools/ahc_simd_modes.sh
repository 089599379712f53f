#!/bin/bash
# ahCluster host time per trial-solver mode (DRFE_AHC_SIMD: 0 scalar, 1 AVX2, 2 AVX-512F) on the host harness
# (drfe_planes_ahc_from_blocks, room frame, no device needed).  Usage: bash tools/ahc_simd_modes.sh
lscpu | grep -E "Model name"
for m in 0 1 2; do
  echo "mode $m"
  DRFE_AHC_SIMD=$m DRFE_TRACE_PLANES=1 python tools/ahc_host_probe.py 2>&1 | grep -E "ahCluster" |
    awk '{for(i=1;i<=NF;i++) if($i=="ahCluster") {c+=$(i+1); k++}} END {print "ahCluster", c/k, "ms"}'
done
