#!/bin/bash
# kernel trace of the matrix-core DDC for c2 and c3
cd /tmp && export TMPDIR=/tmp
export GSDR_DDC_MFMA=1
for w in c2; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_mfma_$w
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --workload $w --no-extras --no-cpu > $OUT/log.txt 2>&1 || true
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $w"; head -8 "$f"
done
