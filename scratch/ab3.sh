#!/bin/bash
run() { wl=$1; name=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps 200 --warmup 10 --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl $name', d['value'], 'Msps  ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])
"
}
for i in 1 2; do
for wl in c2 c3; do
  run $wl prev GSDR_LIB=$PWD/scratch/libgsdr_prev.so
  run $wl new X=1
done; done
