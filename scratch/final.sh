#!/bin/bash
# milestone run: smoke, full GPU test suite, default bench (+ extras + cpu baseline)
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -5 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py > gpurun_out/bench_default.log 2>&1
rc=$?
tail -3 gpurun_out/bench_default.log | cut -c1-1800
echo bench_rc=$rc
