#!/bin/bash
# timing-only builds of the ring loop (WRONG results): which work the C3 time and clock depend on.
# Builds libgsdr variants on the GPU box itself (hipcc is there), runs C3 for ~4 s each and
# samples rocm-smi meanwhile.
set -e
cp gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h /tmp/ring_gen_saved.h
for ab in ${ABLATIONS:-none rot prod lds gload bar rot,prod rot,prod,lds rot,prod,lds,gload}; do
  if [ $ab = none ]; then cp /tmp/ring_gen_saved.h gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h; else GEN_ABLATE=$ab python3 tools/gen_ddc_mfma_ring.py > gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h; fi
  make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
  python bench.py --workload c3 --api inorder --no-extras --no-cpu --steps 20000 --warmup 50 > /tmp/b.json 2>/dev/null &
  BP=$!
  sleep 3.2
  PW=$(rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/.*: //' | tr '\n' ' ')
  wait $BP
  python -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('ablate=%-14s' % '$ab', 'ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'], '| $PW')"
done
cp /tmp/ring_gen_saved.h gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h
make -C gpu_sdr_amd/csrc > /tmp/make.log 2>&1
