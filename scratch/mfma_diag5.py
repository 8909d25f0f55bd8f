"""Constant input, short windows: which k-step is wrong, and what replaced it?"""
import os, sys
import numpy as np
import torch
import gpu_sdr_amd as g

N, L, rate = 256, 1_000_000, 200_000_000
dev = torch.device("cuda:0")
for M, F in [(32, 1), (64, 1), (128, 1), (40, 4)]:
    Lb = (L // M) * M
    freq = np.zeros(N, dtype=np.int64)
    def make(mfma):
        os.environ["GSDR_DDC_MFMA"] = "1" if mfma else "0"
        p = g.param(mode="RX", rate=rate, buffer_len=Lb, decim=M, pf_average=F,
                    freq=[int(v) for v in freq], wave_type=[g.w_type.DIRECT] * N)
        return g.RX_buffer_demodulator(p, device_index=0)
    a, b = make(False), make(True)
    h = np.asarray(a.window(), dtype=np.float64)
    MF = M * F
    nk = (MF + 31) // 32 * 4
    hp = np.concatenate([np.zeros(128), h, np.zeros(8 * nk - MF + 160)])
    tot = h.sum()
    x = torch.from_numpy(np.ones(Lb, np.complex64)).to(dev)
    oa = torch.empty(a.out_capacity, dtype=torch.complex64, device=dev)
    ob = torch.empty(b.out_capacity, dtype=torch.complex64, device=dev)
    seen = {}
    for c in range(4):
        na = a.process(x, oa); nb = b.process(x, ob)
        torch.cuda.synchronize()
        ya = oa[:na].reshape(-1, N).cpu().numpy(); yb = ob[:nb].reshape(-1, N).cpu().numpy()
        r = (yb / ya).real - 1.0
        bad = np.abs(r) > 3e-5
        for v in np.round(r[bad], 5):
            seen[float(v)] = seen.get(float(v), 0) + 1
    print(f"M={M} F={F} blocks={nk // 4}: bad value -> count", dict(sorted(seen.items(), key=lambda kv: -kv[1])[:8]))
    for v in list(seen)[:6]:
        m = []
        for k in range(nk):
            s0 = hp[128 + 8 * k: 136 + 8 * k].sum()
            if abs(-s0 / tot - v) < 2.5e-5:
                m.append((k, "missing"))
            for sh in (-96, -64, -32, -24, -16, -8, 8, 16, 24, 32, 64, 96):
                s1 = hp[128 + 8 * k + sh: 136 + 8 * k + sh].sum()
                if abs((s1 - s0) / tot - v) < 2.5e-5:
                    m.append((k, sh))
        print("   ", v, "matches (step k, shift or missing):", m[:12])
    a.close(); b.close()
