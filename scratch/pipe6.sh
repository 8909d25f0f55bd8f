#!/bin/bash
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pipe6_tests.log 2>&1
rc=$?
tail -3 gpurun_out/pipe6_tests.log
[ $rc -ne 0 ] && exit $rc
run() {
  timeout -k 10 120 python bench.py --workload $WL --api $1 --no-extras --no-cpu --steps 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL asm=$GSDR_MFMA_ASM $1 value', d['value'], 'ms', d['ms_per_step'], 'kernel_us', (d['roofline'] or {}).get('kernel_us'))
" || exit 1
}
for WL in c2 c3; do
export GSDR_MFMA_ASM=2; run inorder; run pipelined
export GSDR_MFMA_ASM=3; run inorder
done
