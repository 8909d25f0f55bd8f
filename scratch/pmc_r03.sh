#!/bin/bash
# usage: scratch/pmc_r03.sh <workload> <tag> [env assignments...]  -- PMC passes of the in-order entry (counters of their own runs, no tracing)
WL=$1; TAG=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="$R/bench.py ${BENCH_EXTRA:-} --workload $WL --api inorder --no-extras --no-cpu --no-host-api --steps 30 --warmup 5 --min-seconds 0.01"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1 || true
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("gsdr::ddc") or k.startswith("gsdr::absmax") or k.startswith("gsdr::chirp") or "pfb_" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, d in out.items():
    print(k, json.dumps(d))
PY
