#!/bin/bash
# same-box A/B: round 2's library against the current one, C2 and C3, interleaved
for rep in 1 2 3; do for lib in r02 cur; do for wl in c2 c3; do
  if [ $lib = r02 ]; then export GSDR_LIB=$PWD/scratch/libgsdr_r02.so GSDR_LIB_OLD_ABI=1; EX=--ablation; else unset GSDR_LIB GSDR_LIB_OLD_ABI; EX=; fi
  python bench.py $EX --workload $wl --no-extras --no-cpu --no-host-api --steps 200 --warmup 20 --min-seconds 1.0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rep$rep %-4s $wl pipelined us/step %7.2f  inorder %7.2f  kernel_us %7.2f %s' % ('$lib', d['ms_per_step']*1e3, d['inorder']['ms_per_step']*1e3, r['kernel_us'], r['kernel']))"
done; done; done
