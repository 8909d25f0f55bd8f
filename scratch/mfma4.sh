#!/bin/bash
set -o pipefail
export PYTHONPATH=$PWD
export GSDR_MFMA_ASM=2
timeout -k 5 100 python scratch/mfma_diag.py 256 100 2>&1 | grep buffer || exit 1
timeout -k 5 100 python scratch/mfma_diag.py 2048 1000 2>&1 | grep buffer || exit 1
timeout -k 5 200 python scratch/mfma_diag5.py 2>&1 | grep "M=" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or engine or direct or pfb or noise or golden or c2 or c3" > gpurun_out/mfma_tests.log 2>&1
rc=$?
tail -4 gpurun_out/mfma_tests.log
[ $rc -ne 0 ] && exit $rc
for w in c2 c3 pfb; do
  for asm in 0 2; do
    GSDR_DDC_MFMA=1 GSDR_MFMA_ASM=$asm timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 5 --no-extras --no-cpu > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    grep -h '^{' gpurun_out/ab.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$w asm=$asm', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_us'])
"
  done
done
