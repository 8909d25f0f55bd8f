#!/bin/bash
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden" > gpurun_out/r03_pfb_pytest2.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r03_pfb_pytest2.log
python scratch/pfb_sweep.py 1024 1230 1016 1292 202 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_pfb_sweep_cu2.log
STAMP_PY=scratch/stamp_pfb3.py bash scratch/stamp_pfb.sh 1230 1016 2>&1 | tee gpurun_out/r03_stamp_pfb_cu2.log
