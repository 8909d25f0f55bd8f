#!/bin/bash
# same-box A/B: butterflies of a short stage on the low halves of all waves (GSDR_PFB_SPREAD=1) against the first waves
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise" > gpurun_out/r03_pytest_spread0.log 2>&1; echo "pytest (spread 0) rc=$?"
GSDR_PFB_SPREAD=1 python -m pytest tests -m gpu -q -x -k "pfb or tones or noise" > gpurun_out/r03_pytest_spread1.log 2>&1; echo "pytest (spread 1) rc=$?"; tail -2 gpurun_out/r03_pytest_spread1.log
for rep in 1 2; do
echo "== shipped, rep $rep"; python scratch/pfb_sweep.py 64 256 512 1024 2048 4096 2>&1 | grep TONES
echo "== GSDR_PFB_SPREAD=1, rep $rep"; GSDR_PFB_SPREAD=1 python scratch/pfb_sweep.py 64 256 512 1024 2048 4096 2>&1 | grep TONES
done | tee gpurun_out/r03_pfb_ab_spread.log
