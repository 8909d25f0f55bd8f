#!/bin/bash
# builds a stamp build of the library next to the shipped one and runs the PFB probe with it
set -e
make -C gpu_sdr_amd/csrc OUT=$PWD/scratch/libgsdr_stamp.so SERVER=/tmp/none_server RXLINK=/tmp/none_rxlink FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -w -DGSDR_STAMP_BUILD" $PWD/scratch/libgsdr_stamp.so > /tmp/stamp_make.log 2>&1 || { tail -20 /tmp/stamp_make.log; exit 1; }
for n in ${@:-1024 1230 64}; do
  GSDR_LIB=$PWD/scratch/libgsdr_stamp.so python ${STAMP_PY:-scratch/stamp_pfb.py} $n 2>&1 | tail -1
done
