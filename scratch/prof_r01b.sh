#!/bin/bash
# round-1 final profiles: default bench command per workload (pipelined entry for the DDC),
# plus an in-order kernel trace for the kernel-alone durations
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for WL in c2 c3; do
  OUT=$R/gpurun_out/prof_${WL}_io
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --workload $WL --api inorder --no-extras --no-cpu > $OUT/trace.log 2>&1 || true
  echo "inorder trace $WL done"
done
cd $R
for WL in c2 c3 pfb c4; do
  bash scratch/prof.sh $WL $WL > gpurun_out/prof_$WL.log 2>&1
  echo "prof $WL done"
done
