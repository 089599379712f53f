#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02 [batch] [extra bench.py arguments, e.g. "--config 5"]
# one --kernel-trace --stats pass and four counter passes (FETCH_SIZE | WRITE_SIZE | two SQ groups), never combined with
# tracing; everything lands under gpurun_out/<tag>_*; tools/pmc_traffic.py folds the counter CSVs into profiles/.
set -u
TAG=${1:-r02}
B=${2:-512}
EXTRA=${3:-}
rm -rf ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_stats ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_pmc_*      # one run per directory: the fold picks the only CSV
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
CMD="python3 $R/bench.py --steps 5 --warmup 2 --batch $B --inflight 1 --distinct 8 --render-workers 1 --preload 0 --no-cpu-baseline --no-extras $EXTRA"
# the timing pass runs long enough to leave the device's clock transient behind (200 single-context steps before 20 timed ones);
# the counter passes stay short: counters do not depend on the clock and serialise the kernels
CMD_STATS="python3 $R/bench.py --steps 20 --warmup 5 --batch $B --inflight 1 --distinct 8 --render-workers 1 --preload 200 --no-cpu-baseline --no-extras $EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $CMD_STATS > $OUT/${TAG}_stats.log 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
            "sq1 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
            "sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    set -- $pass
    name=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- $CMD > $OUT/${TAG}_pmc_$name.log 2>&1
done
find $OUT -name "${TAG}_*" -name "*.csv" | head -20
