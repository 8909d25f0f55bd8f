#!/bin/bash
# end-of-milestone GPU job: tests, default bench, rocprof evidence for C2 and C3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke_rc=$? >> gpurun_out/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_rc=$? >> gpurun_out/pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1; echo bench_rc=$? >> gpurun_out/bench_default.log
timeout -k 10 300 python bench.py --workload c3 --steps 200 --warmup 10 --no-extras > gpurun_out/bench_c3.log 2>&1
bash scratch/prof.sh c2 c2 > gpurun_out/prof_c2.log 2>&1
bash scratch/prof.sh c3 c3 > gpurun_out/prof_c3.log 2>&1
bash scratch/prof.sh c4 c4 > gpurun_out/prof_c4.log 2>&1
tail -2 gpurun_out/smoke.log; tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench_default.log | cut -c1-3000
