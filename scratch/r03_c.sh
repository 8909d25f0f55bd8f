#!/bin/bash
# suite (stop at first failure) + in-order kernel traces of C2 / C3
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest4.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r03_pytest4.log
cd /tmp && export TMPDIR=/tmp
for w in c3 c2; do
  for api in inorder pipelined; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_${w}_$api -o $w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --api $api --steps 200 --warmup 20 --min-seconds 0.2 --no-extras --no-cpu --no-host-api > $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_${w}_$api.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_${w}_$api.err
  echo "$w $api rc=$?"
  done
done
