#!/bin/bash
# PMC counters of the MFMA kernels on c3 (and c2)
cd /tmp && export TMPDIR=/tmp
export GSDR_DDC_MFMA=1
for w in c3; do
for asm in 1; do
  export GSDR_MFMA_ASM=$asm
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma_${w}_asm$asm
  mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --workload $w --no-extras --no-cpu > $OUT/p1.log 2>&1 || true
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --workload $w --no-extras --no-cpu > $OUT/p2.log 2>&1 || true
done
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for d in sorted(glob.glob(root + "/pmc_mfma_*")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ddc_mfma" in k:
                agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(os.path.basename(d), k)
        print("   ", {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
