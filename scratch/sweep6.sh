#!/bin/bash
run() { wl=$1; name=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps 300 --warmup 10 --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl $name', d['value'], 'Msps  ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'])
"
}
for n in 143 167 192 250 333; do run c3 nch$n GSDR_DDC_NCH=$n; done
for n in 1112 1250 1429 1536 1667 2000; do run c2 nch$n GSDR_DDC_NCH=$n; done
run c3 auto X=1
run c2 auto X=1
