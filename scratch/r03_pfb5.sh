#!/bin/bash
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pfb_pytest5.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_pfb_pytest5.log
echo "== library's choice, radix 8 / 6 / 10"; python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 4096 2>&1 | grep -v amdgpu.ids | grep TONES | tee gpurun_out/r03_pfb_sweep_r8.log
echo "== radix 4 / 2 as before"; GSDR_PFB_RADIX8=0 python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 2>&1 | grep -v amdgpu.ids | grep TONES | tee -a gpurun_out/r03_pfb_sweep_r8.log
echo "== run kernel forced, radix 8"; GSDR_PFB_CU=1 python scratch/pfb_sweep.py 64 256 1024 2048 2>&1 | grep -v amdgpu.ids | grep TONES | tee -a gpurun_out/r03_pfb_sweep_r8.log
