#!/bin/bash
run() { wl=$1; name=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps 300 --warmup 10 --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl $name', d['value'], 'Msps  ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'])
"
}
for n in 1000 1250 1600 2000 2500; do run c2 nch$n GSDR_DDC_NCH=$n; done
for n in 125 200 250; do run c3 nch$n GSDR_DDC_NCH=$n; done
