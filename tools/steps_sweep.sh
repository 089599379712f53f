#!/bin/bash
# ms per step of the headline loop against the number of timed steps (the driver times 20): what the fixed cost of a timed region is
for k in 20 20 40 100 300; do
  python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('steps', $k, 'value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'timed ms', round(d['ms_per_step']*$k,2))
"
done
