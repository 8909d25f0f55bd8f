#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16p" > gpurun_out/r02_prec_pytest.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r02_prec_pytest.log
