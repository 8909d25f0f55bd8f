"""Where does the matrix-core DDC differ from the packed-FP32 kernel?  (GPU)"""
import os, sys
import numpy as np
import torch
import gpu_sdr_amd as g
from gpu_sdr_amd.source import device_tones, tone_comb

N, M, F, L, rate = int(sys.argv[1]), int(sys.argv[2]), 4, int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000, 200_000_000
dev = torch.device("cuda:0")
freq, ampl, phase = tone_comb(N, rate, seed=20251004)
def make(mfma):
    os.environ["GSDR_DDC_MFMA"] = "1" if mfma else "0"
    p = g.param(mode="RX", rate=rate, buffer_len=L, decim=M, pf_average=F,
                freq=[int(v) for v in freq], wave_type=[g.w_type.DIRECT] * N)
    return g.RX_buffer_demodulator(p, device_index=0)
a, b = make(False), make(True)
x = torch.empty(L, dtype=torch.complex64, device=dev)
oa = torch.empty(a.out_capacity, dtype=torch.complex64, device=dev)
ob = torch.empty(b.out_capacity, dtype=torch.complex64, device=dev)
for c in range(4):
    device_tones(x, rate - L - 12345 + c * L, rate, freq, ampl, phase, sigma=1e-3, seed=77 + c)
    na = a.process(x, oa); nb = b.process(x, ob)
    torch.cuda.synchronize()
    ya = oa[:na].reshape(-1, N); yb = ob[:nb].reshape(-1, N)
    d = (ya - yb).abs()
    scale = ya.abs().mean()
    bad = (d > 1e-4 * scale).nonzero()
    print("buffer", c, "max diff/scale %.3e" % float(d.max() / scale), "bad elements", len(bad))
    if len(bad):
        bm = (d > 1e-4 * scale)
        print("  bad fraction per tone tile of 32:", [round(float(bm[:, t*32:(t+1)*32].float().mean()), 3) for t in range(min(N // 32, 16))])
        pr = bm.float().mean(dim=1)
        print("  bad fraction per row %% 32:", [round(float(pr[k::32].mean()), 2) for k in range(32)])
        print("  bad fraction per row tile (first 12, last 4):", [round(float(pr[k*32:(k+1)*32].mean()), 2) for k in list(range(12)) + list(range(len(pr)//32 - 4, len(pr)//32))])
        rows = bad[:, 0].cpu().numpy(); cols = bad[:, 1].cpu().numpy()
        if c == 1:
            for o in np.unique(rows)[:5]:
                cc = cols[rows == o]
                print("  row", o, "row%32", o % 32, "tones", cc.min(), "..", cc.max(), "n", len(cc))
                print("    flat:", ya[o, cc[:4]].cpu().numpy())
                print("    mfma:", yb[o, cc[:4]].cpu().numpy())
                print("    mfma/flat:", (yb[o, cc[:4]] / ya[o, cc[:4]]).cpu().numpy())
                if o + 1 < ya.shape[0]:
                    print("    flat next row:", ya[o + 1, cc[:4]].cpu().numpy())
        print("  rows", np.unique(rows)[:40], "... tiles", np.unique(rows // 32)[:40])
        print("  tones", np.unique(cols)[:64])
