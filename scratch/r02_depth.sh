#!/bin/bash
for rep in 1 2; do for depth in 2 3 4; do for streams in 2 3; do
  GSDR_BENCH_DEPTH=$depth GSDR_PIPE_STREAMS=$streams python bench.py --workload c3 --no-extras --no-cpu --no-host-api --steps 200 --warmup 20 --min-seconds 1.0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep$rep depth=$depth streams=$streams Msps %8.1f us/step %7.2f' % (d['value'], d['ms_per_step']*1e3))"
done; done; done
