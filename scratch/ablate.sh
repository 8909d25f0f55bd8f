#!/bin/bash
export GSDR_DDC_MFMA=1 GSDR_MFMA_ASM=1
for w in c3 c2; do
for n in full loads mfma rot_conv_loads rot_conv_loads_mfma; do
  if [ $n = full ]; then unset GSDR_LIB; else export GSDR_LIB=$PWD/scratch/libgsdr_ab_$n.so; fi
  timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-extras --no-cpu 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$w ablate=$n', d['roofline']['kernel'], d['roofline']['kernel_us'])
"
done
done
