"""Impulse comb input: for each bad output, which tap did the kernel apply instead of h[t0]?"""
import os, sys
import numpy as np
import torch
import gpu_sdr_amd as g
import collections

N, M, F, L, rate = 256, 100, 4, 1_000_000, 200_000_000
dev = torch.device("cuda:0")
freq = np.zeros(N, dtype=np.int64)
def make(mfma):
    os.environ["GSDR_DDC_MFMA"] = "1" if mfma else "0"
    p = g.param(mode="RX", rate=rate, buffer_len=L, decim=M, pf_average=F,
                freq=[int(v) for v in freq], wave_type=[g.w_type.DIRECT] * N)
    return g.RX_buffer_demodulator(p, device_index=0)
a, b = make(False), make(True)
h = np.asarray(a.window(), dtype=np.float64)
stats = collections.Counter()
for c in range(0, 1000, 37):
    xh = np.zeros(L, np.complex64); xh[c::1000] = 1.0
    x = torch.from_numpy(xh).to(dev)
    oa = torch.empty(a.out_capacity, dtype=torch.complex64, device=dev)
    ob = torch.empty(b.out_capacity, dtype=torch.complex64, device=dev)
    for rep in range(2):
        na = a.process(x, oa); nb = b.process(x, ob)
        torch.cuda.synchronize()
    ya = oa[:na].reshape(-1, N).cpu().numpy().real; yb = ob[:nb].reshape(-1, N).cpu().numpy().real
    bad = np.argwhere(np.abs(ya - yb) > 1e-7)
    rows = np.unique(bad[:, 0])
    for o in rows[:4000]:
        # impulse positions seen by row o: n = c + 1000 j, t = n - (o-3)*100 in [0,400)
        start = (o - 3) * 100
        ts = [t for t in range(400) if (start + t - c) % 1000 == 0 and 0 <= start + t < L]
        if len(ts) != 1:
            continue
        t0 = ts[0]
        val = yb[o, 0]
        cand = [t for t in range(max(0, t0 - 128), min(400, t0 + 129)) if abs(h[t] - val) < 2e-7]
        if abs(val) < 1e-9:
            stats[(t0 // 8, "zero")] += 1
        elif cand:
            d = min(cand, key=lambda t: abs(t - t0)) - t0
            stats[(t0 // 8, d)] += 1
        else:
            stats[(t0 // 8, "other %.3g vs %.3g" % (val, h[t0]))] += 1
print("(step k of the impulse, applied-tap offset): count")
for k, v in sorted(stats.items(), key=lambda kv: -kv[1])[:40]:
    print(k, v)
