"""Constant input, non-zero tones: complex ratio mfma/flat of the bad elements."""
import os, sys
import numpy as np
import torch
import gpu_sdr_amd as g

N, M, F, L, rate = 256, 100, 4, 1_000_000, 200_000_000
dev = torch.device("cuda:0")
freq = (np.arange(N) - 128) * 100_000 + 12_345
def make(mfma):
    os.environ["GSDR_DDC_MFMA"] = "1" if mfma else "0"
    p = g.param(mode="RX", rate=rate, buffer_len=L, decim=M, pf_average=F,
                freq=[int(v) for v in freq], wave_type=[g.w_type.DIRECT] * N)
    return g.RX_buffer_demodulator(p, device_index=0)
a, b = make(False), make(True)
x = torch.from_numpy(np.ones(L, np.complex64)).to(dev)
oa = torch.empty(a.out_capacity, dtype=torch.complex64, device=dev)
ob = torch.empty(b.out_capacity, dtype=torch.complex64, device=dev)
np.set_printoptions(precision=4, linewidth=220, suppress=True)
for c in range(3):
    na = a.process(x, oa); nb = b.process(x, ob)
    torch.cuda.synchronize()
    ya = oa[:na].reshape(-1, N).cpu().numpy().astype(np.complex128); yb = ob[:nb].reshape(-1, N).cpu().numpy().astype(np.complex128)
    scale = np.abs(ya).max(axis=0, keepdims=True)        # per tone
    d = np.abs(yb - ya) / scale
    bad = np.argwhere(d > 1e-4)
    print("buffer", c, "bad", len(bad))
    if len(bad) and c == 1:
        rows = np.unique(bad[:, 0])
        for o in rows[:6]:
            cols = bad[bad[:, 0] == o][:, 1]
            print(" row", o, "(row%32 =", o % 32, ") tones", cols.min(), "..", cols.max(), "count", len(cols))
            n = cols[0]
            print("   mfma/flat at bad tones:", (yb[o, cols[:6]] / ya[o, cols[:6]]))
            print("   (mfma-flat)/scale     :", ((yb[o, cols[:6]] - ya[o, cols[:6]]) / scale[0, cols[:6]]))
