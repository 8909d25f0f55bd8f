#!/bin/bash
# the bench line and the rocprofv3 --kernel-trace --stats summary of the SAME command on the SAME box
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/samebox
CMD="$R/bench.py --workload c3 --api inorder --no-extras --no-cpu --no-host-api --steps 200 --warmup 20"
python3 $CMD > $R/gpurun_out/samebox/c3_inorder_line.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/samebox/trace -- python3 $CMD > $R/gpurun_out/samebox/c3_inorder_line_traced.json 2>/dev/null
python3 $R/bench.py --no-cpu > $R/gpurun_out/samebox/default_line.json 2>/dev/null
find $R/gpurun_out/samebox -name "*kernel_stats.csv" | head -2
