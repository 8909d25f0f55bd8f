#!/bin/bash
# LDS bank-conflict counters of the TONES kernel (separate --pmc pass, dispatches serialised)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lds; rm -rf $OUT; mkdir -p $OUT
P="$R/bench.py --workload pfb --api inorder --no-extras --no-cpu --no-host-api --steps 30 --warmup 5 --min-seconds 0.01"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 $P > $OUT/p1.log 2>&1
echo "rc=$?"; tail -3 $OUT/p1.log
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmc_lds/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row.get("Kernel_Name",""); 
        if "pfb" in k: acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
