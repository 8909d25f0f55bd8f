#!/bin/bash
# quick A/B of chunking / occupancy settings on C2 and C3
for wl in c3 c2; do
for w in 3 4 5 6; do
  GSDR_DDC_WAVES_PER_SIMD=$w timeout -k 10 120 python bench.py --steps 100 --warmup 5 --workload $wl --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl wps=$w', d['value'], 'Msps  kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])
"
done
done
