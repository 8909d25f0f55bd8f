#!/bin/bash
# same-box ablation of the C2 kernel's new pieces (timing only for the variants)
for rep in 1 2 3; do for lib in r02 ldsse noepi; do for wl in c2; do
  if [ $lib = r02 ]; then export GSDR_LIB=$PWD/scratch/libgsdr_r02.so GSDR_LIB_OLD_ABI=1; else export GSDR_LIB=$PWD/scratch/lib_abl_$lib.so; unset GSDR_LIB_OLD_ABI; fi
  python bench.py --ablation --workload $wl --no-extras --no-cpu --no-host-api --steps 200 --warmup 20 --min-seconds 0.7 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rep$rep %-9s $wl pipelined us/step %7.2f  inorder %7.2f  kernel_us %7.2f %s' % ('$lib', d['ms_per_step']*1e3, d['inorder']['ms_per_step']*1e3, r['kernel_us'], r['kernel']))"
done; done; done
