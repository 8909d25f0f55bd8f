#!/bin/bash
# timing-only builds (WRONG results) against the C2 time per buffer, in order and overlapped
set -e
cp gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h /tmp/ring_gen_saved.h
for ab in ${ABLATIONS:-none bimg}; do
  if [ $ab = none ]; then cp /tmp/ring_gen_saved.h gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h; else GEN_ABLATE=$ab python3 tools/gen_ddc_mfma_ring.py > gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h; fi
  make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
  for api in inorder pipelined; do
  python bench.py --workload c2 --api $api --no-extras --no-cpu --steps 3000 --warmup 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate=%-8s %-9s' % ('$ab','$api'), 'us/buffer', round(d['ms_per_step']*1e3,2))"
  done
done
cp /tmp/ring_gen_saved.h gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h
make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1
