"""Constant input at full size: magnitude of the errors of bad elements (ratio mfma/flat - 1)."""
import os, sys
import numpy as np
import torch
import gpu_sdr_amd as g

N, M, F, L, rate = 256, 100, 4, 1_000_000, 200_000_000
dev = torch.device("cuda:0")
freq = np.zeros(N, dtype=np.int64)
def make(mfma):
    os.environ["GSDR_DDC_MFMA"] = "1" if mfma else "0"
    p = g.param(mode="RX", rate=rate, buffer_len=L, decim=M, pf_average=F,
                freq=[int(v) for v in freq], wave_type=[g.w_type.DIRECT] * N)
    return g.RX_buffer_demodulator(p, device_index=0)
a, b = make(False), make(True)
x = torch.from_numpy(np.ones(L, np.complex64)).to(dev)
oa = torch.empty(a.out_capacity, dtype=torch.complex64, device=dev)
ob = torch.empty(b.out_capacity, dtype=torch.complex64, device=dev)
for c in range(3):
    na = a.process(x, oa); nb = b.process(x, ob)
    torch.cuda.synchronize()
    ya = oa[:na].reshape(-1, N).cpu().numpy(); yb = ob[:nb].reshape(-1, N).cpu().numpy()
    r = (yb / ya).real - 1.0
    bad = np.abs(r) > 1e-4
    print("buffer", c, "bad", int(bad.sum()))
    if bad.any():
        vals = np.round(r[bad], 4)
        u, cnt = np.unique(vals, return_counts=True)
        order = np.argsort(-cnt)[:12]
        print("  most common ratio-1 values:", [(float(u[i]), int(cnt[i])) for i in order])
