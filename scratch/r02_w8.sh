#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "w8" > gpurun_out/r02_w8_pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r02_w8_pytest.log
for asm in ${ASMS:-4 5}; do for wl in ${WLS:-c3 c2}; do
  GSDR_MFMA_ASM=$asm timeout -k 10 120 python bench.py --workload $wl --no-extras --no-cpu --steps 200 --warmup 20 --min-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('asm=$asm $wl value', d['value'], 'us/step', round(d['ms_per_step']*1e3,2), 'inorder', round(d['inorder']['ms_per_step']*1e3,2), 'kernel_us', r['kernel_us'], r['kernel'])"
done; done
