#!/bin/bash
# A/B of generator variants on ONE box, interleaved: scratch/ab.sh <gen script> <header> <asm kind> "<name>=<ENV=val ...>" ...
# builds one library per variant into scratch/, then runs the variants alternately (boxes and
# thermal state drift by a few per cent: never compare across runs).
GEN=$1; HDR=$2; ASM=$3; shift 3
set -e
cp $HDR /tmp/ab_saved.h
names=()
for v in "$@"; do
  name=${v%%=*}; envs=${v#*=}
  env $envs python3 $GEN > $HDR
  make -C gpu_sdr_amd/csrc OUT=$PWD/scratch/lib_ab_$name.so SERVER=/tmp/none_s RXLINK=/tmp/none_r $PWD/scratch/lib_ab_$name.so > /tmp/make_$name.log 2>&1 || { tail -5 /tmp/make_$name.log; exit 1; }
  names+=($name)
done
cp /tmp/ab_saved.h $HDR
for rep in 1 2 3; do for name in "${names[@]}"; do for wl in ${WLS:-c3}; do
  GSDR_MFMA_ASM=$ASM GSDR_LIB=$PWD/scratch/lib_ab_$name.so python bench.py --ablation ${BENCH_EXTRA:-} --workload $wl --no-extras --no-cpu --steps 200 --warmup 20 --min-seconds ${SECS:-1.5} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rep$rep %-10s $wl pipelined us/step %7.2f  inorder %7.2f  kernel_us %7.2f' % ('$name', d['ms_per_step']*1e3, d['inorder']['ms_per_step']*1e3, r['kernel_us']))"
done; done; done
