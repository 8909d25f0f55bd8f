#!/bin/bash
# priority variants of the x16 loop: in-order C3 kernel time + WG end-time split
set -e
cp gpu_sdr_amd/csrc/ddc_mfma_ring16_gen.h /tmp/ring16_saved.h
for pr in ${PRIOS:-none ab}; do
  GEN_PRIO=$pr python3 tools/gen_ddc_mfma_ring16.py > gpu_sdr_amd/csrc/ddc_mfma_ring16_gen.h
  make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
  for wl in ${WLS:-c3 c2}; do
  GSDR_MFMA_ASM=4 python bench.py --workload $wl --no-extras --no-cpu --steps 200 --warmup 20 --min-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('prio=$pr $wl value', d['value'], 'us/step', round(d['ms_per_step']*1e3,2), 'inorder', round(d['inorder']['ms_per_step']*1e3,2), 'kernel_us', r['kernel_us'])"
  done
done
cp /tmp/ring16_saved.h gpu_sdr_amd/csrc/ddc_mfma_ring16_gen.h
make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1
