#!/bin/bash
# usage: scratch/prof.sh <workload> <tag>
set -e
WL=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --workload $WL --no-extras --no-cpu > $OUT/trace.log 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --workload $WL --no-extras --no-cpu > $OUT/pmc1.log 2>&1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --workload $WL --no-extras --no-cpu > $OUT/pmc2.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc5 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --workload $WL --no-extras --no-cpu > $OUT/pmc5.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --workload $WL --no-extras --no-cpu > $OUT/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --workload $WL --no-extras --no-cpu > $OUT/pmc4.log 2>&1 || true
find $OUT -name "*.csv" | head -30
