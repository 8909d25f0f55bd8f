#!/bin/bash
# timing-only builds of the packed-FP32 VALU engine (WRONG results): what the scalar loads and their waits cost at C3
# ABLATE=1: no scalar loads, no waits; =2: loads issued, never waited for; (none): the shipped kernel
set -e
# built here: make -C gpu_sdr_amd/csrc OUT=scratch/libgsdr_abN.so FLAGS="... -DGSDR_TIMING_BUILD -DGSDR_ABLATE=N"
# (the two libraries are built beforehand: see the FLAGS in the git history of this file; -DGSDR_TIMING_BUILD -DGSDR_ABLATE=1|2)
run() { python bench.py --workload c3 --api inorder --no-extras --no-cpu --no-host-api --steps 100 --warmup 10 --min-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s us/step %7.2f  kernel_us %7.2f %s' % ('$1', d['ms_per_step']*1e3, r['kernel_us'], r['kernel']))"; }
for rep in 1 2; do
  GSDR_DDC_MFMA=0 run "shipped"
  GSDR_DDC_MFMA=0 GSDR_LIB=$PWD/scratch/libgsdr_ab1.so run "no loads, no waits" --ablation
  GSDR_DDC_MFMA=0 GSDR_LIB=$PWD/scratch/libgsdr_ab2.so run "loads, no waits" --ablation
done
