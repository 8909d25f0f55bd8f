#!/bin/bash
# compute streams x outstanding buffers of the pipelined entry
for WL in c2 pfb c3; do
for cfg in "2 3" "2 4" "3 3" "3 4"; do
  set -- $cfg
  GSDR_PIPE_STREAMS=$1 GSDR_BENCH_DEPTH=$2 python bench.py --workload $WL --api pipelined --no-extras --no-cpu --steps 600 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL streams=$1 depth=$2 ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'])"
done; done
