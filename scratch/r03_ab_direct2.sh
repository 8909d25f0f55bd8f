#!/bin/bash
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_direct.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03_pytest_direct.log
for rep in 1 2; do
echo "== GSDR_PFB_CU=1: everything through the run kernel (direct filter), rep $rep"; GSDR_PFB_CU=1 python scratch/pfb_sweep.py 16 64 100 256 512 1024 2048 2>&1 | grep "TONES\|NOISE"
echo "== the library's choice, rep $rep"; python scratch/pfb_sweep.py 16 64 100 256 512 1024 2048 2>&1 | grep "TONES\|NOISE"
done | tee gpurun_out/r03_pfb_ab_direct2.log
