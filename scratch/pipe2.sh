#!/bin/bash
run() {
  timeout -k 10 120 python bench.py --api pipelined --no-extras --no-cpu --steps 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 value', d['value'], 'ms', d['ms_per_step'], 'kernel_us', (d['roofline'] or {}).get('kernel_us'), 'inorder', (d.get('inorder') or {}).get('ms_per_step'))
" || exit 1
}
GSDR_PIPE_PRIO=0 run "prio=0 hwq=default"
GSDR_PIPE_PRIO=1 run "prio=1 hwq=default"
GSDR_PIPE_PRIO=0 GPU_MAX_HW_QUEUES=8 run "prio=0 hwq=8"
GSDR_PIPE_PRIO=0 GPU_MAX_HW_QUEUES=16 run "prio=0 hwq=16"
GSDR_PIPE_PRIO=1 GPU_MAX_HW_QUEUES=8 run "prio=1 hwq=8"
