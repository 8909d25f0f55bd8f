"""Diagnostic: phases of the workgroups of pfb_cu_kernel (stamp build, scratch/stamp_pfb.sh builds the library).
Slots: 0 start, 1 raw samples in the LDS, 2 filter done, 3 / 4 after stages 0 / 1, 5 after the last stage, 7 end."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g
from gpu_sdr_amd import _lib
dbg = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
nfft = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = min(1024, nfft)
rng = np.random.default_rng(nfft)
freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=N, replace=False)]
p = g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=4, fft_tones=nfft, freq=freq, wave_type=[g.w_type.TONES] * N)
dem = g.RX_buffer_demodulator(p, device_index=0)
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev)
for k in range(200):
    dem.process_device(x[k % 4], out)
torch.cuda.synchronize()
stamps = torch.zeros(8 * 20000, dtype=torch.int64, device=dev)
dbg.gsdr_debug_set_fft_stamp_buffer.argtypes = [C.c_void_p]
assert dbg.gsdr_debug_set_fft_stamp_buffer(C.c_void_p(stamps.data_ptr())) == 0
for rep in range(2):
    stamps.zero_()
    torch.cuda.synchronize()
    dem.process_device(x[0], out)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[(s[:, 0] > 0) & (s[:, 7] > 0)]
    t = s * 0.01
    base = t[:, 0].min()
    names = {1: "raw_in_lds", 2: "filter", 3: "stage0", 4: "stage1", 5: "rest_of_stages", 7: "output"}
    d = dict(kernel=dem.kernel_name, nfft=nfft, wgs=len(s), start_spread_us=round(float(t[:, 0].max() - base), 2),
             last_end_us=round(float(t[:, 7].max() - base), 2), mean_life_us=round(float((t[:, 7] - t[:, 0]).mean()), 2))
    prev = 0
    for sl in (1, 2, 3, 4, 5, 7):
        if s[:, sl].max() > 0:
            d[names[sl] + "_us"] = round(float((t[:, sl] - t[:, prev]).mean()), 2)
            prev = sl
    if s[:, 6].max() > 0:
        d["stage0_until_mfma_loop_done_us"] = round(float((t[:, 6] - t[:, 2]).mean()), 2)
    print(json.dumps(d))
dbg.gsdr_debug_set_fft_stamp_buffer(None)
dem.close()
