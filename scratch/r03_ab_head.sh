#!/bin/bash
# same-box A/B of a library change: scratch/libgsdr_head.so (the previous commit's build) against the working tree's
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_ab_head.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_ab_head.log
for rep in 1 2; do
echo "== before, rep $rep"; GSDR_LIB=$PWD/scratch/libgsdr_head.so timeout -k 10 120 python scratch/pfb_sweep.py ${SIZES:-64 256 1000 1024 1230 1016 2048 4096 8192} 2>&1 | grep "TONES"
echo "== after, rep $rep"; timeout -k 10 120 python scratch/pfb_sweep.py ${SIZES:-64 256 1000 1024 1230 1016 2048 4096 8192} 2>&1 | grep "TONES"
done | tee gpurun_out/r03_pfb_ab_head.log
