#!/bin/bash
# same-box A/B: the library before the 24-bit multiplies (5e99384) against the shipped one
for rep in 1 2; do
echo "== before (5e99384), rep $rep"; GSDR_LIB=$PWD/scratch/libgsdr_pre24.so python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 2>&1 | grep TONES
echo "== shipped, rep $rep"; python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 2>&1 | grep TONES
done | tee gpurun_out/r03_pfb_ab_mul24.log
