#!/bin/bash
# the packed-FP32 VALU engine at C3: sub-blocks of 20 (default) and 40 samples, waves per SIMD
for rep in 1 2; do for v in "K=20" "K=40" "K=40 GSDR_DDC_WAVES_PER_SIMD=4" "K=20 GSDR_DDC_WAVES_PER_SIMD=8"; do
  set -- $v; k=${1#K=}; shift
  env GSDR_DDC_MFMA=0 GSDR_DDC_K=$k "$@" python bench.py --workload c3 --api inorder --no-extras --no-cpu --no-host-api --steps 100 --warmup 10 --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rep$rep %-40s us/step %7.2f  kernel_us %7.2f %s frac %.4f' % ('$v', d['ms_per_step']*1e3, r['kernel_us'], r['kernel'], r['frac']))"
done; done
