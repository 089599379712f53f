#!/bin/bash
# headline step at the driver's protocol (--steps 20 --warmup 5) for several batch sizes: frames/s and ms per step
for b in 512 512 1024 2048; do
  python bench.py --steps 20 --warmup 5 --batch $b --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch', $b, 'value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))
"
done
