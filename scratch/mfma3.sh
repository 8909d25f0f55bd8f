#!/bin/bash
set -o pipefail
export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or engine or direct or pfb or noise or golden or c2 or c3" > gpurun_out/mfma_tests.log 2>&1
rc=$?
tail -4 gpurun_out/mfma_tests.log
[ $rc -ne 0 ] && exit $rc
for w in c2 c3; do
  for cfg in "1 0" "1 1"; do
    set -- $cfg
    GSDR_DDC_MFMA=$1 GSDR_MFMA_ASM=$2 timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 5 --no-extras --no-cpu > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    grep -h '^{' gpurun_out/ab.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$w mfma=$1 asm=$2', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_us'])
"
  done
done
