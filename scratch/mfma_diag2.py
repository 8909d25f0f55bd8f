"""What do the wrong rows contain?  Constant input, per-row ratio mfma/flat for a few tones."""
import os, sys
import numpy as np
import torch
import gpu_sdr_amd as g

N, M, F, L, rate = 64, 100, 4, 100_000, 200_000_000
dev = torch.device("cuda:0")
freq = np.zeros(N, dtype=np.int64); freq[1] = 1_000_000; freq[2] = -37_000_000
def make(mfma):
    os.environ["GSDR_DDC_MFMA"] = "1" if mfma else "0"
    p = g.param(mode="RX", rate=rate, buffer_len=L, decim=M, pf_average=F,
                freq=[int(v) for v in freq], wave_type=[g.w_type.DIRECT] * N)
    return g.RX_buffer_demodulator(p, device_index=0)
a, b = make(False), make(True)
mode = sys.argv[1] if len(sys.argv) > 1 else "const"
if mode == "const":
    xh = np.ones(L, np.complex64)
else:  # one sample position inside each 32-sample block: which samples are lost?
    xh = ((np.arange(L) % 32) == int(mode)).astype(np.complex64)
x = torch.from_numpy(xh).to(dev)
oa = torch.empty(a.out_capacity, dtype=torch.complex64, device=dev)
ob = torch.empty(b.out_capacity, dtype=torch.complex64, device=dev)
for c in range(2):
    na = a.process(x, oa); nb = b.process(x, ob)
    torch.cuda.synchronize()
ya = oa[:na].reshape(-1, N).cpu().numpy(); yb = ob[:nb].reshape(-1, N).cpu().numpy()
np.set_printoptions(precision=4, linewidth=200, suppress=True)
for t in (0, 1):
    r = yb[32:96, t] / ya[32:96, t]
    print("tone", t, "ratio rows 32..95 (re):"); print(r.real.reshape(4, 16)); print("(im):"); print(r.imag.reshape(4, 16))
