#!/bin/bash
set -e
cp gpu_sdr_amd/csrc/ddc_mfma_ring16w8_gen.h /tmp/w8_saved.h
for pr in ${PRIOS:-none turns}; do
  GEN_PRIO=$pr python3 tools/gen_ddc_mfma_ring16w8.py > gpu_sdr_amd/csrc/ddc_mfma_ring16w8_gen.h
  make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
  GSDR_MFMA_ASM=5 python bench.py --workload c3 --no-extras --no-cpu --steps 200 --warmup 20 --min-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('prio=$pr c3 value', d['value'], 'us/step', round(d['ms_per_step']*1e3,2), 'inorder', round(d['inorder']['ms_per_step']*1e3,2), 'kernel_us', r['kernel_us'])"
done
cp /tmp/w8_saved.h gpu_sdr_amd/csrc/ddc_mfma_ring16w8_gen.h
make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1
