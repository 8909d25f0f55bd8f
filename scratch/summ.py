import csv, sys, glob, collections, os
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    print("==", f)
    for r in list(csv.DictReader(open(f)))[:8]:
        print({k: r[k] for k in r if k in ("Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs")})
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    print("==", f)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "ddc" in k or "chirp" in k:
            print(k, {c: (sum(v)/len(v), len(v)) for c, v in d.items()})
