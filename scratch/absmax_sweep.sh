#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for c in 1024 2048 4096 8192 16384 32768; do
  export GSDR_ABSMAX_CHUNK=$c
  OUT=$GRAFT_REPO_ROOT/gpurun_out/absmax_$c
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --workload c2 --no-extras --no-cpu > $OUT/log.txt 2>&1 || true
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "chunk $c: $(grep absmax "$f" | awk -F, '{print $4}') ns avg;  bench: $(grep -o '"ms_per_step": [0-9.]*' $OUT/log.txt)"
done
