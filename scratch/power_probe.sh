#!/bin/bash
# power and clocks of the GPU while a workload streams for several seconds
rocm-smi --showmaxpower 2>&1 | grep -i "Max Graphics" | head -1
for cfg in "c3 40000" "c2 200000" "pfb 300000" "c4 1500000"; do
  set -- $cfg
  python bench.py --workload $1 --no-extras --no-host-api --no-cpu --steps $2 --warmup 100 > /tmp/b_$1.json 2>/dev/null &
  BP=$!
  sleep 3.5
  for i in 1 2; do
    rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.7
  done
  wait $BP
  python -c "
import json
d=json.loads(open('/tmp/b_$1.json').read().strip().splitlines()[-1]); print('$1', d['value'], 'Msamples/s,', d['ms_per_step']*1e3, 'us per buffer')"
done
