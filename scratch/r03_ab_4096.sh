#!/bin/bash
python -m pytest tests -m gpu -q -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_4096.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r03_pytest_4096.log
for rep in 1 2; do
echo "== the library's choice, rep $rep"; python scratch/pfb_sweep.py 2560 3000 3072 4000 4096 2>&1 | grep "TONES\|NOISE"
echo "== GSDR_PFB_CU=0, rep $rep"; GSDR_PFB_CU=0 python scratch/pfb_sweep.py 2560 3000 3072 4000 4096 2>&1 | grep "TONES\|NOISE"
done | tee gpurun_out/r03_pfb_ab_4096.log
