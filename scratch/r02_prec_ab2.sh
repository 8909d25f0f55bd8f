#!/bin/bash
for rep in 1 2; do for prec in 0 1; do for wl in c2 pfb; do
  GSDR_MFMA_PREC=$prec python bench.py --workload $wl --no-extras --no-cpu --no-host-api --steps 200 --warmup 20 --min-seconds 1.0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rep$rep prec=$prec $wl Msps %8.1f pipelined us/step %9.2f  inorder %9.2f  kernel_us %9.2f %s' % (d['value'], d['ms_per_step']*1e3, d['inorder']['ms_per_step']*1e3, r['kernel_us'], r['kernel']))"
done; done; done
