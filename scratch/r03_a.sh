#!/bin/bash
# round 3, first checkpoint: the whole GPU suite, then the default bench line
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r03_pytest2.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r03_pytest2.log
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r03_bench_a.json; tail -3 gpurun_out/r03_bench_a.err
