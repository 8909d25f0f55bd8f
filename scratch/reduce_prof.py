"""Reduces gpurun_out/prof_<wl>/ and prof_<wl>_io/ (scratch/prof_r03.sh) to the files committed under profiles/."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
TAG = os.environ.get("PROF_TAG", "r03")


def kname(full):
    """'void gsdr::(anonymous namespace)::pfb_lds_kernel<true>(...)' -> 'gsdr::pfb_lds_kernel<true>'"""
    n = full.replace("(anonymous namespace)::", "")
    n = n[5:] if n.startswith("void ") else n
    return n.split("(")[0]


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def kernel_stats(src_dir, dst):
    f = newest(os.path.join(src_dir, "trace", "**", "*kernel_stats.csv"))
    if not f:
        return None
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if r and kname(r[0]).startswith("gsdr::")]
    with open(dst, "w", newline="") as o:
        csv.writer(o, quoting=csv.QUOTE_ALL).writerows(keep)
    return {kname(r[0]): float(r[3]) / 1e3 for r in keep[1:]}


def overlap(src_dir):
    """average number of gsdr:: kernels resident, and the steady-state period per main launch"""
    f = newest(os.path.join(src_dir, "trace", "**", "*kernel_trace.csv"))
    if not f:
        return None
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))
          if r["Kernel_Name"].startswith("gsdr::ddc") or r["Kernel_Name"].startswith("gsdr::absmax")]
    main = sorted(e for e in ev if "ddc" in e[2])
    main = main[len(main) // 3:]
    if len(main) < 4:
        return None
    t0, t1 = main[0][0], main[-1][1]
    busy = sum(min(e[1], t1) - max(e[0], t0) for e in ev if e[1] > t0 and e[0] < t1)
    return dict(main_launches=len(main), period_us=round((main[-1][0] - main[0][0]) / (len(main) - 1) / 1e3, 2),
                avg_kernels_resident=round(busy / (t1 - t0), 2))


def pmc(src_dir, dst):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src_dir, "pmc*"))):
        if not os.path.isdir(d):
            continue
        f = newest(os.path.join(d, "**", "*counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k.startswith("gsdr::") and "source" not in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in acc.items()}
    json.dump(out, open(dst, "w"), indent=1)
    return out


def dominant(stats):
    """the gsdr:: DDC / chirp kernel with the largest share of the in-order run"""
    best = None
    for k, v in (stats or {}).items():
        if ("ddc_mfma" in k or "chirp" in k or "ddc_flat" in k or "pfb_lds" in k or "pfb_cu" in k) and "convert" not in k:
            if best is None or v[1] > best[1]:
                best = (k, v[1])
    return best[0] if best else None


def kernel_stats2(src_dir, dst):
    """like kernel_stats, but returns {kernel: (avg_us, total_ns)}"""
    f = newest(os.path.join(src_dir, "trace", "**", "*kernel_stats.csv"))
    if not f:
        return None
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if r and kname(r[0]).startswith("gsdr::")]
    with open(dst, "w", newline="") as o:
        csv.writer(o, quoting=csv.QUOTE_ALL).writerows(keep)
    return {kname(r[0]): (float(r[3]) / 1e3, float(r[2])) for r in keep[1:]}


traffic = {}
for wl in ("c2", "c3", "c3flat", "pfb", "c4"):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + wl)
    if not os.path.isdir(src):
        continue
    st = kernel_stats2(src, os.path.join(OUT, f"{TAG}_{wl}_kernel_stats.csv"))
    pm = pmc(src, os.path.join(OUT, f"{TAG}_{wl}_pmc.json"))
    io = os.path.join(ROOT, "gpurun_out", "prof_" + wl + "_io")
    st_io = kernel_stats2(io, os.path.join(OUT, f"{TAG}_{wl}_inorder_kernel_stats.csv")) if os.path.isdir(io) else None
    k = dominant(st_io) or dominant(st)
    c = pm.get(k, {})
    entry = {"kernel_inorder": k, "rocprof_avg_us_inorder": round(st_io[k][0], 2) if st_io and k in st_io else None,
             "overlapped_entry_kernels_avg_us": {kk: round(v[0], 2) for kk, v in (st or {}).items()
                                                 if "ddc" in kk or "chirp" in kk or "absmax" in kk or "pfb" in kk}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        entry.update(FETCH_SIZE_KB_raw=c["FETCH_SIZE"], WRITE_SIZE_KB_raw=c["WRITE_SIZE"],
                     hbm_bytes_per_launch=int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
                     note="counters of the in-order kernel (dispatches serialised: kernel alone on the chip); FETCH_SIZE doubled per "
                          "MI355X_MICROARCH.md (gfx950 reports 1/2 of streamed reads); scalar-load and 8-B-per-lane store widths are "
                          "uncalibrated, so treat as +-2x")
    traffic[wl] = entry
    print(wl, json.dumps(entry)[:400])
json.dump(traffic, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)
