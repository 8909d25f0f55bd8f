#!/bin/bash
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 5 --workload $WL --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$WL $name', d['value'], 'Msps  kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])
"
}
for WL in c3 c2; do
run normal X=1
run abl1_noloads GSDR_LIB=$PWD/scratch/libgsdr_abl1.so
run abl2_nowait GSDR_LIB=$PWD/scratch/libgsdr_abl2.so
done
