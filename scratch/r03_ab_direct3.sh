#!/bin/bash
python -m pytest tests -m gpu -q -k "pfb or tones or noise or golden or fuzz or server or rxlink or pipelined" > gpurun_out/r03_pytest_direct.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_pytest_direct.log
for rep in 1 2; do
echo "== the library's choice, rep $rep"; python scratch/pfb_sweep.py 64 128 256 512 1000 1024 1230 1016 2048 4096 2>&1 | grep "TONES"
echo "== GSDR_PFB_DIRECT=0, rep $rep"; GSDR_PFB_DIRECT=0 python scratch/pfb_sweep.py 128 256 1024 1230 2048 2>&1 | grep "TONES"
done | tee gpurun_out/r03_pfb_ab_direct3.log
