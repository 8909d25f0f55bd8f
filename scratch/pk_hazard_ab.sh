#!/bin/bash
# Is rule R3 a hazard ACROSS kernels?  scratch/concurrent_handles.py with the shipped library (no packed FP32
# in chirp / mix / generic DDC / FFT kernels) and with a variant that lets the compiler use v_pk_*_f32 there.
set -e
make -C gpu_sdr_amd/csrc OUT=$PWD/scratch/libgsdr_pk.so SERVER=/tmp/none_server RXLINK=/tmp/none_rxlink FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -w -DGSDR_NO_PK=" $PWD/scratch/libgsdr_pk.so > /tmp/pk_make.log 2>&1 || { tail -20 /tmp/pk_make.log; exit 1; }
for rep in 1 2; do
  echo "== shipped library"
  timeout -k 10 150 python scratch/concurrent_handles.py ${1:-15} 2>&1 | tail -3
  echo "== variant with packed FP32 in the small kernels"
  GSDR_LIB=$PWD/scratch/libgsdr_pk.so timeout -k 10 150 python scratch/concurrent_handles.py ${1:-15} 2>&1 | tail -6
done
