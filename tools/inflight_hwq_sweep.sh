#!/bin/bash
# headline loop: batches in flight x hardware queues (GPU_MAX_HW_QUEUES), 200 timed steps each
for q in 4 8; do for n in 3 4 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 200 --warmup 5 --inflight $n --no-extras --no-cpu-baseline --preload 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('hwq', $q, 'inflight', $n, 'value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3))
"
done; done
