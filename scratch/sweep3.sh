#!/bin/bash
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 5 --workload $WL --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$WL $name', d['value'], 'Msps  kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])
"
}
WL=c3
for n in 200 250 300 333; do run nch$n GSDR_DDC_NCH=$n; done
run nch250_k16 GSDR_DDC_NCH=250 GSDR_DDC_K=16
run nch333_k16 GSDR_DDC_NCH=333 GSDR_DDC_K=16
WL=c2
for n in 2000 2500 3333; do run nch$n GSDR_DDC_NCH=$n; done
run nch2500_k16 GSDR_DDC_NCH=2500 GSDR_DDC_K=16
