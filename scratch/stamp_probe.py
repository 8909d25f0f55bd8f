"""Diagnostic: where inside a C3 launch do the workgroups start and end?  Needs the stamp build
(scratch/stamp_probe.sh builds it: -DGSDR_STAMP_BUILD).  Prints, per launch, the spread of start
and end times over the workgroups and per XCD."""
import ctypes as C, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpu_sdr_amd import _lib
L = _lib.lib()
dbg = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
eng = bench.HipEngine()
dem, bufs, outs, N = eng.build(wl, dev, 20251004)
nwg = 4096
stamps = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
st = torch.cuda.Stream(dev)
for k in range(3000):     # reach the power-capped steady state first
    dem.process_device(bufs[k % 8], outs[0], st)
torch.cuda.synchronize()
dbg.gsdr_debug_set_stamp_buffer.argtypes = [C.c_void_p]
assert dbg.gsdr_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr())) == 0
res = []
for k in range(6):
    stamps.zero_()
    torch.cuda.synchronize()
    for j in range(20):
        dem.process_device(bufs[j % 8], outs[0], st)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] > 0]
    t0, t1, xcc, hwid = s[:, 0] * 0.01, s[:, 1] * 0.01, s[:, 2] & 0xf, s[:, 2] >> 8      # us
    base = t0.min()
    d = dict(wgs=len(s), start_spread_us=round(float(t0.max() - base), 2), first_end_us=round(float(t1.min() - base), 2),
             last_end_us=round(float(t1.max() - base), 2), mean_life_us=round(float((t1 - t0).mean()), 2),
             min_life_us=round(float((t1 - t0).min()), 2), max_life_us=round(float((t1 - t0).max()), 2),
             per_xcd_mean_end_us={int(x): round(float((t1[xcc == x] - base).mean()), 1) for x in np.unique(xcc)},
             per_xcd_mean_life_us={int(x): round(float((t1 - t0)[xcc == x].mean()), 1) for x in np.unique(xcc)})
    res.append(d)
    if k == 5:
        # workgroups that shared a compute unit: same XCC, same (se, sh, cu) bits of HW_ID
        cu = (hwid >> 8) & 0xff
        life = t1 - t0
        import collections
        groups = collections.defaultdict(list)
        for i in range(len(s)):
            groups[(int(xcc[i]), int(cu[i]))].append((round(float(t1[i] - base), 1)))
        sizes = collections.Counter(len(v) for v in groups.values())
        print("groups by (xcc, HW_ID[15:8]):", dict(sizes))
        pairs = [sorted(v) for v in groups.values() if len(v) == 2]
        if pairs:
            a = np.array(pairs)
            print("pairs on one CU: first end mean %.1f, second end mean %.1f; spread of first %.1f..%.1f, of second %.1f..%.1f" % (
                a[:, 0].mean(), a[:, 1].mean(), a[:, 0].min(), a[:, 0].max(), a[:, 1].min(), a[:, 1].max()))
        print("hw_id samples:", [hex(int(h)) for h in hwid[:8]])
        wid = hwid & 0xf
        print("wave_id values:", dict(collections.Counter(int(w) for w in wid)))
        g2 = collections.defaultdict(list)
        for i in range(len(s)):
            g2[(int(xcc[i]), int(cu[i]))].append((float(t1[i]), int(wid[i]), int((hwid[i] >> 4) & 3)))
        first_w = collections.Counter(sorted(v)[0][1] for v in g2.values() if len(v) == 2)
        print("wave_id of the workgroup that ends first:", dict(first_w))
        print("end-time histogram (us):", np.histogram(t1 - base, bins=10)[0].tolist(), np.round(np.histogram(t1 - base, bins=10)[1], 1).tolist())
    if s[:, 3].max() > 0:      # slot 3: end of the assembly block (ring16 kernel)
        t3 = s[:, 3] * 0.01
        late = t0 - base > 2.0      # workgroups of the second round
        d["asm_end_minus_start_us"] = dict(round1=round(float((t3 - t0)[~late].mean()), 2), round2=round(float((t3 - t0)[late].mean()), 2) if late.any() else None)
        d["end_minus_asm_end_us"] = dict(round1=round(float((t1 - t3)[~late].mean()), 2), round2=round(float((t1 - t3)[late].mean()), 2) if late.any() else None)
        if s[:, 4].max() > 0:  # slot 4: just before the assembly block
            t4 = s[:, 4] * 0.01
            d["asm_start_minus_start_us"] = dict(round1=round(float((t4 - t0)[~late].mean()), 2), round2=round(float((t4 - t0)[late].mean()), 2) if late.any() else None)
        d["round2_wgs"] = int(late.sum())
        d["round2_start_us"] = [round(float((t0 - base)[late].min()), 1), round(float((t0 - base)[late].max()), 1)] if late.any() else None
    print(json.dumps(d))
dbg.gsdr_debug_set_stamp_buffer(None)
dem.close()
