#!/bin/bash
# kernel trace + counters of the frame-per-workgroup PFB kernel at one frame length: scratch/prof_pfb_size.sh 1024
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=${1:-1024}; O=$R/gpurun_out/prof_pfbsz_$N
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/scratch/pfb_sweep.py $N > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 $R/scratch/pfb_sweep.py $N > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $O/pmc2 -- python3 $R/scratch/pfb_sweep.py $N > $O/pmc2.log 2>&1
python3 - <<P
import csv,glob,collections
f=glob.glob("$O/trace/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pfb_lds" in r["Name"]: print("kernel avg us", float(r["AverageNs"])/1e3, "min", float(r["MinNs"])/1e3, "calls", r["Calls"])
acc=collections.defaultdict(list)
for f in glob.glob("$O/pmc*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pfb_lds" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v),1) for k,v in sorted(acc.items())})
P
