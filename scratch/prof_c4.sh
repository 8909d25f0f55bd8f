#!/bin/bash
# kernel trace of the C4 bench command: is the 4 us step the kernel or the host's launch rate?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_c4q
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4q/trace -- python3 $R/bench.py --workload c4 --no-extras --no-cpu --no-host-api --steps 200 --warmup 20 --min-seconds 0.2 > $R/gpurun_out/prof_c4q/trace.log 2>&1
f=$(find $R/gpurun_out/prof_c4q/trace -name "*kernel_stats.csv" | head -1)
cut -c1-60,200-400 $f | head -5
python3 - <<P
import csv,glob
f=glob.glob("$R/gpurun_out/prof_c4q/trace/**/*kernel_trace.csv",recursive=True)[0]
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "chirp_lockin" in r["Kernel_Name"]]
ev.sort(); ev=ev[len(ev)//2:]
import statistics
print("launches", len(ev), "avg duration us", statistics.mean(e[1]-e[0] for e in ev)/1e3, "avg period us", (ev[-1][0]-ev[0][0])/(len(ev)-1)/1e3,
      "avg gap us", statistics.mean(ev[i+1][0]-ev[i][1] for i in range(len(ev)-1))/1e3)
P
