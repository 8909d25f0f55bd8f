#!/bin/bash
# which HIP call of the first submit()s is the slow one: HIP API trace of scratch/worst_call_probe.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/worst_trace; rm -rf $O; mkdir -p $O
rocprofv3 --hip-trace --output-format csv -d $O -- python3 $R/scratch/worst_call_probe.py > $O/run.log 2>&1
tail -4 $O/run.log
python3 - <<P
import csv, glob
f = glob.glob("$O/**/*hip_api_trace.csv", recursive=True)[0]
rows = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"])) for r in csv.DictReader(open(f))]
t_first_submit = None
rows.sort(key=lambda r: r[2])
# the slow calls inside the loop: everything after the last hipHostMalloc of the set-up
big = [r for r in rows if r[0] > 1_000_000]
for d, fn, ts in big[-40:]:
    print(round(d / 1e6, 3), "ms", fn, "at", round((ts - rows[0][2]) / 1e9, 3), "s")
P
