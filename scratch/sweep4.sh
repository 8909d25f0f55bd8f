#!/bin/bash
run() { wl=$1; name=$2; shift 2
  env "$@" timeout -k 10 120 python bench.py --steps 200 --warmup 10 --workload $wl --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl $name', d['value'], 'Msps  kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])
"
}
for WL in c3 c2; do
run $WL normal GSDR_DDC_AUTOTUNE=0
run $WL abl1_noloads GSDR_DDC_AUTOTUNE=0 GSDR_LIB=$PWD/scratch/libgsdr_abl1.so
run $WL abl2_nowait GSDR_DDC_AUTOTUNE=0 GSDR_LIB=$PWD/scratch/libgsdr_abl2.so
run $WL normal_nopf GSDR_DDC_AUTOTUNE=0 GSDR_DDC_PREFETCH=0
done
