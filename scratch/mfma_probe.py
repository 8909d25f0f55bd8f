"""Debug build of the assembly loop: dump producer/consumer registers (constant input,
one-block window) and report which lanes/dwords deviate from the majority.

Method record: the dump hooks (GEN_DEBUG in the generator, MfmaLaunch::dbg, gsdrx_debug_read)
were removed from the product after the hunt; they are in git history (commit da7983e and
its successors up to the "single load round trip" commit)."""
import os, ctypes as C
import numpy as np
import torch
os.environ["GSDR_DDC_MFMA"] = "1"; os.environ["GSDR_MFMA_SGB"] = "9"
os.environ["GSDR_MFMA_DEBUG_BYTES"] = str(64 << 20)
import gpu_sdr_amd as g
from gpu_sdr_amd import _lib
N, M, F, rate = 256, 32, 1, 200_000_000
L = 1_000_000 // M * M
dev = torch.device("cuda:0")
p = g.param(mode="RX", rate=rate, buffer_len=L, decim=M, pf_average=F, freq=[0] * N, wave_type=[g.w_type.DIRECT] * N)
b = g.RX_buffer_demodulator(p, device_index=0)
x = torch.from_numpy(np.ones(L, np.complex64)).to(dev)
ob = torch.empty(b.out_capacity, dtype=torch.complex64, device=dev)
for c in range(2):
    b.process(x, ob); torch.cuda.synchronize()
lib = _lib.lib()
lib.gsdrx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
nwg = 8 * ((L // M + 31) // 32 + 7) // 8 * 2
buf = np.empty(nwg * 4 * 8192 // 4, np.uint32)
assert lib.gsdrx_debug_read(b._h, buf.ctypes.data, buf.nbytes) == 0
rec = buf.reshape(nwg * 4, 8, 64, 4)     # [wave][record][lane][dword]
names = ["HI4 (producer)", "F0h", "F1h", "F2h", "F3h", "LO4 (producer)", "XB", "HV"]
valid = rec[:, 0, 0, 0] != 0xffffffff
print("waves with data:", int(valid.sum()), "of", len(valid))
for r in range(8):
    d = rec[valid, r]                     # [wave, lane, dword]
    # majority per (wave-in-wg for producer records, lane-half, dword)
    bad_total = 0
    report = {}
    for wv in range(4):
        dw = d[wv::4] if r in (0, 5, 6, 7) else d
        if r not in (0, 5, 6, 7) and wv:
            break
        for hh in range(2):
            sub = dw[:, hh * 32:(hh + 1) * 32, :]          # rows of one lane half: equal by construction
            ref = np.median(sub.reshape(-1, 4), axis=0).astype(np.uint32) if r >= 5 else None
            vals, counts = np.unique(sub.reshape(-1, 4), axis=0, return_counts=True)
            maj = vals[np.argmax(counts)]
            dev_mask = (sub != maj)
            if dev_mask.any():
                w, l, k = np.nonzero(dev_mask)
                for ll, kk in zip(l, k):
                    report[(wv, hh, int(ll), int(kk))] = report.get((wv, hh, int(ll), int(kk)), 0) + 1
                bad_total += int(dev_mask.sum())
    print(names[r], "deviating dwords:", bad_total)
    if report:
        lanes = sorted({(k[1] * 32 + k[2]) for k in report})
        dws = sorted({k[3] for k in report})
        print("   lanes", lanes[:40], "dwords", dws)
