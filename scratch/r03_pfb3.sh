#!/bin/bash
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz or pipelined or two_front" > gpurun_out/r03_pfb_pytest3.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_pfb_pytest3.log
python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 1292 202 1018 1004 4096 8192 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_pfb_sweep_cu3.log
