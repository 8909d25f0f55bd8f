#!/bin/bash
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/rt_tests.log 2>&1 || { tail -15 gpurun_out/rt_tests.log; exit 1; }
tail -2 gpurun_out/rt_tests.log
for WL in c2 pfb c3; do
for rt in 1 2; do
for api in inorder pipelined; do
  GSDR_MFMA_RT=$rt python bench.py --workload $WL --api $api --no-extras --no-cpu --steps 2000 --warmup 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL rt=$rt %-9s' % '$api', 'us/buffer', round(d['ms_per_step']*1e3,2))"
done; done; done
