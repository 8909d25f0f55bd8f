#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma_g3" 2>&1 | tail -8
run() {
  timeout -k 10 120 python bench.py --workload $WL --api inorder --no-extras --no-cpu --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL asm=$GSDR_MFMA_ASM value', d['value'], 'ms', d['ms_per_step'], d['roofline']['kernel'], 'kernel_us', (d['roofline'] or {}).get('kernel_us'))
" || exit 1
}
for WL in c3 c2 pfb; do
for k in 2 0 4; do export GSDR_MFMA_ASM=$k; run; done
done
