#!/bin/bash
# round 3: suite + in-order kernel traces of C2 / C3 (staging pass duration) + bench line
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest3.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r03_pytest3.log
cd /tmp && export TMPDIR=/tmp
for w in c3 c2 c4 pfb; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_$w -o $w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --api inorder --steps 200 --warmup 20 --min-seconds 0.2 --no-extras --no-cpu --no-host-api > $GRAFT_REPO_ROOT/gpurun_out/prof_r03_$w.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03_$w.err
  echo "$w rc=$?"
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r03_$w -name "*kernel_stats.csv" | head -1); head -6 "$f"
done
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err; echo "bench rc=$?"
