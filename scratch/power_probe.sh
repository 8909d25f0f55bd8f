#!/bin/bash
# power and clocks of the GPU while C3 runs for several seconds
(rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -i "power\|sclk\|mclk" | head -12) 
for WL in c3 c2; do
  python bench.py --workload $WL --api pipelined --no-extras --no-cpu --steps 40000 --warmup 100 > /tmp/b_$WL.json 2>/dev/null &
  BP=$!
  sleep 3.5
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk" | head -4
    sleep 0.7
  done
  wait $BP
  python -c "
import json
d=json.loads(open('/tmp/b_$WL.json').read().strip().splitlines()[-1]); print('$WL', d['value'], d['ms_per_step'])"
done
