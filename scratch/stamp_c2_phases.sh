#!/bin/bash
# C2: lifetimes of the workgroups of an in-order launch, full kernel and the timing-only builds
# (one block only / no stores): where prologue, loop and epilogue spend their time
set -e
make -C gpu_sdr_amd/csrc OUT=$PWD/scratch/libgsdr_stamp.so SERVER=/tmp/none_server RXLINK=/tmp/none_rxlink FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -w -DGSDR_STAMP_BUILD -DGSDR_TIMING_BUILD" $PWD/scratch/libgsdr_stamp.so > /tmp/stamp_make.log 2>&1 || { tail -20 /tmp/stamp_make.log; exit 1; }
for tm in 0 2 1 3; do
  echo "== GSDR_MFMA_TIMING=$tm (1 = no stores, 2 = one block only)"
  GSDR_MFMA_TIMING=$tm GSDR_MFMA_ASM=4 GSDR_LIB=$PWD/scratch/libgsdr_stamp.so python scratch/stamp_probe.py ${1:-c2} 2>&1 | tail -1
done
