#!/bin/bash
# round 2, first GPU call: full GPU suite (margins -> gpurun_out/parity_margins.json), default bench
# line, two-rank rehearsal on the one device (gloo control plane)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02_pytest.log
tail -5 gpurun_out/r02_pytest.log
timeout -k 10 500 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r02_bench_default.err
timeout -k 10 200 python bench.py --gpus 2 --steps 100 --warmup 10 --no-extras --no-cpu > gpurun_out/r02_n2_one_device.json 2> gpurun_out/r02_n2_one_device.err; echo "n2 rc=$?"
tail -c 400 gpurun_out/r02_n2_one_device.err
cat gpurun_out/r02_n2_one_device.json | cut -c1-600
