#!/bin/bash
# full GPU suite, then A/B bench of the staged (2) and single-launch (3) engines
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/direct_tests.log 2>&1
rc=$?
tail -5 gpurun_out/direct_tests.log
[ $rc -ne 0 ] && exit $rc
for k in 2 3; do
  GSDR_MFMA_ASM=$k timeout -k 10 200 python bench.py --steps 300 --warmup 30 > gpurun_out/direct_bench_$k.json 2>gpurun_out/direct_bench_$k.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/direct_bench_$k.json").read().strip().splitlines()[-1])
print("asm=$k", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_us"], "c3", d["extras"]["c3"]["kernel_us"], d["extras"]["c3"]["msamples_per_s"])
PY
done
