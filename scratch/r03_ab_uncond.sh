#!/bin/bash
# same-box A/B: the direct filter's loads guarded per block (01a20cc) against unconditional loads from clamped addresses
python -m pytest tests -m gpu -q -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_uncond.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_uncond.log
for rep in 1 2; do
echo "== before (01a20cc), rep $rep"; GSDR_LIB=$PWD/scratch/libgsdr_prebar.so python scratch/pfb_sweep.py 128 256 512 1000 1024 1230 1016 1536 2048 2>&1 | grep "TONES"
echo "== unconditional loads, rep $rep"; python scratch/pfb_sweep.py 128 256 512 1000 1024 1230 1016 1536 2048 2>&1 | grep "TONES"
done | tee gpurun_out/r03_pfb_ab_uncond.log
