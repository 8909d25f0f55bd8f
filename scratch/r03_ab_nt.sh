#!/bin/bash
python -m pytest tests -m gpu -q -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pytest_nt.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_nt.log
for rep in 1 2; do
echo "== shipped (two workgroups of 512 threads per unit where the direct filter takes them), rep $rep"; python scratch/pfb_sweep.py 128 256 512 1000 1024 1230 1016 1536 2048 2>&1 | grep "TONES"
echo "== GSDR_PFB_CU_NT=1024 (one workgroup of 1024), rep $rep"; GSDR_PFB_CU_NT=1024 python scratch/pfb_sweep.py 128 256 512 1000 1024 1230 1016 1536 2048 2>&1 | grep "TONES"
done | tee gpurun_out/r03_pfb_ab_nt.log
