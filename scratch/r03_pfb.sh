#!/bin/bash
# PFB: parity of everything TONES / NOISE, then the sweep with the run kernel and with the frame-per-workgroup kernel
python -m pytest tests -m gpu -q -x -k "pfb or tones or noise or golden or fuzz or pipelined" > gpurun_out/r03_pfb_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r03_pfb_pytest.log
python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 1018 1004 4096 2>&1 | tee gpurun_out/r03_pfb_sweep_cu.log
GSDR_PFB_CU=0 python scratch/pfb_sweep.py 64 256 1000 1024 1230 2048 1016 2>&1 | tee gpurun_out/r03_pfb_sweep_wg.log
