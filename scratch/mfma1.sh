#!/bin/bash
# GPU run of the matrix-core DDC: parity, then bench A/B over its knobs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or engine or direct or pfb or noise or golden or c2 or c3" > gpurun_out/mfma_tests.log 2>&1
rc=$?
tail -15 gpurun_out/mfma_tests.log
[ $rc -ne 0 ] && exit $rc
for w in c2 c3; do
  for cfg in "0 1 4 4" "1 1 4 0" "1 1 4 2" "1 1 4 3" "1 1 4 4" "1 2 4 0" "1 2 4 3" "1 2 4 4" "1 1 2 0"; do
    set -- $cfg
    GSDR_DDC_MFMA=$1 GSDR_MFMA_TT=$2 GSDR_MFMA_W=$3 GSDR_MFMA_SGB=$4 timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-extras --no-cpu > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    grep -h '^{' gpurun_out/ab.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$w mfma=$1 TT=$2 W=$3 SGB=$4', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_us'])
"
  done
done
