#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined or profile_sampling or two_handles or dynamic_range" > gpurun_out/pipe_tests.log 2>&1
rc=$?
tail -15 gpurun_out/pipe_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scratch/pipe_probe.py 2>&1 | tee gpurun_out/pipe_probe.log
