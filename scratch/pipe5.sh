#!/bin/bash
for tm in 0 1 2 3; do
for api in pipelined inorder; do
  GSDR_MFMA_TIMING=$tm timeout -k 10 120 python bench.py --workload c2 --api $api --no-extras --no-cpu --steps 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('timing=$tm $api ms', d['ms_per_step'], 'kernel_us', (d['roofline'] or {}).get('kernel_us'))
" || exit 1
done; done
