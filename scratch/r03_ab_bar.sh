#!/bin/bash
# same-box A/B: __syncthreads() as a call (01a20cc) against the inlined fence + s_barrier + fence
python -m pytest tests -m gpu -q -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_bar.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_bar.log
for rep in 1 2; do
echo "== before (01a20cc: __syncthreads() is a call), rep $rep"; GSDR_LIB=$PWD/scratch/libgsdr_prebar.so python scratch/pfb_sweep.py 64 256 1000 1024 1230 1016 2048 4096 2>&1 | grep "TONES"
echo "== inlined barrier, rep $rep"; python scratch/pfb_sweep.py 64 256 1000 1024 1230 1016 2048 4096 2>&1 | grep "TONES"
done | tee gpurun_out/r03_pfb_ab_bar.log
