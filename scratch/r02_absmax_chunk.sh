#!/bin/bash
# C2/C3 in order: samples per workgroup of the staging / absmax pass (GSDR_ABSMAX_CHUNK)
for ch in 4096 2048 1024 8192; do
  for w in c2 c3; do
    GSDR_ABSMAX_CHUNK=$ch python bench.py --workload $w --api inorder --no-extras --no-cpu --no-host-api --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('chunk $ch', '$w', 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', d['roofline']['kernel_us'])"
  done
done
