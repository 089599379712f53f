#!/bin/bash
# usage: ab.sh libA libB ... : alternates the variants, 3 rounds each, prints value + stage ms
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for l in "$@"; do
    python tools/bench_variant.py build/$l --steps 40 --warmup 5 --no-cpu-baseline --no-extras --distinct 16 2>/dev/null | python -c "
import sys,json
for x in sys.stdin:
    if x.startswith('{'):
        d=json.loads(x); print('$l', round(d['value']), d['stage_ms_per_batch'])
"
  done
done
