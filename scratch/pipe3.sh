#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined" 2>&1 | tail -3 || exit 1
run() {
  timeout -k 10 120 python bench.py --workload $WL --api pipelined --no-extras --no-cpu --steps 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL $1 value', d['value'], 'ms', d['ms_per_step'], 'kernel_us', (d['roofline'] or {}).get('kernel_us'))
" || exit 1
}
for WL in c2 c3; do
GSDR_PIPE_STREAMS=2 GSDR_BENCH_DEPTH=3 run "streams=2 depth=3"
GSDR_PIPE_STREAMS=2 GSDR_BENCH_DEPTH=4 run "streams=2 depth=4"
GSDR_PIPE_STREAMS=3 GSDR_BENCH_DEPTH=3 run "streams=3 depth=3"
GSDR_PIPE_STREAMS=3 GSDR_BENCH_DEPTH=4 run "streams=3 depth=4"
done
