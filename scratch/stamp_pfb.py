"""Diagnostic: phases of the workgroups of pfb_lds_kernel (stamp build, scratch/stamp_pfb.sh).
Slots: 0 start, 1 filter done (first barrier), 2.. after stage 0,1,2,3,(4+), 7 end."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g
from gpu_sdr_amd import _lib
dbg = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
L, rate = int(os.environ.get("STAMP_L", "1000000")), 200_000_000
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
if os.environ.get("STAMP_MASK"):
    assert dbg.gsdr_debug_set_fft_stamp_mask(C.c_uint(int(os.environ["STAMP_MASK"], 0))) == 0
for rep in range(3):
    stamps.zero_()
    torch.cuda.synchronize()
    for k in range(10):
        dem.process_device(x[k % 4], out)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[(s[:, 0] > 0) & (s[:, 7] > 0)]
    t = s * 0.01
    base = t[:, 0].min()
    d = dict(nfft=nfft, wgs=len(s), start_spread_us=round(float(t[:, 0].max() - base), 2), last_end_us=round(float(t[:, 7].max() - base), 2),
             mean_life_us=round(float((t[:, 7] - t[:, 0]).mean()), 2),
             filter_us=round(float((t[:, 1] - t[:, 0]).mean()), 2))
    if s[:, 6].max() > 0 and nfft in (256, 1024, 2048):      # slot 6: core clocks over the workgroup's life (lds kernel, <= 4 stages)
        d["core_clock_ghz"] = round(float((s[:, 6] / ((s[:, 7] - s[:, 0]) * 10.0)).mean()), 3)
        s[:, 6] = 0
    prev = 1
    for sl in range(2, 7):
        if s[:, sl].max() > 0:
            d["stage%d_us" % (sl - 2)] = round(float((t[:, sl] - t[:, prev]).mean()), 2)
            prev = sl
    d["output_us"] = round(float((t[:, 7] - t[:, prev]).mean()), 2)
    print(json.dumps(d))
dbg.gsdr_debug_set_fft_stamp_buffer(None)
dem.close()
