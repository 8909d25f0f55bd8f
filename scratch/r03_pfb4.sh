#!/bin/bash
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz" > gpurun_out/r03_pfb_pytest4.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_pfb_pytest4.log
echo "== library's choice"; python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 4096 8192 16384 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_pfb_sweep_r16.log
echo "== run kernel forced"; GSDR_PFB_CU=1 python scratch/pfb_sweep.py 64 256 1024 2048 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_pfb_sweep_r16.log
