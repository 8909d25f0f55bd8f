#!/bin/bash
# kernel timeline of the C2 step for the staged (2) and the single-launch (3) engine
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/gaps
for k in 2 3; do
  export GSDR_MFMA_ASM=$k
  rm -rf /tmp/gp$k
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp$k -- python3 $R/bench.py --steps 300 --warmup 30 > $R/gpurun_out/gaps/bench_$k.json 2> $R/gpurun_out/gaps/err_$k.log || exit 1
  f=$(find /tmp/gp$k -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $k <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# longest run of consecutive ddc/absmax kernels
seq = [(r["Kernel_Name"].split("(")[0][-28:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
best = []
cur = []
for s in seq:
    if "ddc_mfma" in s[0] or "absmax" in s[0]:
        cur.append(s)
    else:
        if len(cur) > len(best): best = cur
        cur = []
if len(cur) > len(best): best = cur
best = best[len(best)//4:]          # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(best, best[1:]):
    dur[a[0]].append(a[2]-a[1]); gap[a[0] + " -> " + b[0]].append(b[1]-a[2])
print("asm =", sys.argv[2], "kernels in run:", len(best))
for k, v in dur.items(): print("  dur  %-40s %8.2f us (n=%d)" % (k, sum(v)/len(v)/1e3, len(v)))
for k, v in gap.items(): print("  gap  %-60s %8.2f us" % (k, sum(v)/len(v)/1e3))
print("  period %.2f us" % ((best[-1][1]-best[0][1]) / 1e3 / (len(best)-1) * (2 if sys.argv[2]=="2" else 1)))
PY
done
