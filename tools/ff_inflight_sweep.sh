for cfg in "3 2" "4 2" "4 4" "5 4" "6 4"; do set -- $cfg; echo "inflight $1 cape $2"; DRFE_FF_INFLIGHT=$1 DRFE_FF_CAPE=$2 python tools/full_frontend_sweep.py 512 2>&1 | grep -v amdgpu; done
