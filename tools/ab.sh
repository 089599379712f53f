#!/bin/bash
# A/B of variant builds inside ONE gpurun call (boxes differ by 2-3 %): tools/ab.sh libA.so libB.so ...   (files under build/)
# alternates the variants three times with the bench's defaults (three batches in flight, 1000 steps); AB_ARGS overrides them
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${AB_ARGS:-}
for r in 1 2 3; do
  for l in "$@"; do
    python tools/bench_variant.py build/$l --no-cpu-baseline --no-extras --distinct 16 $ARGS 2>/dev/null | python -c "
import sys,json
for x in sys.stdin:
    if x.startswith('{'):
        d=json.loads(x); print('$l', round(d['value']), d['stage_ms_per_batch'])
"
  done
done
