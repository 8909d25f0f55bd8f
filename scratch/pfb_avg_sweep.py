"""TONES per 1 M-sample buffer over pf_average (taps per bin) at two frame lengths."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g
dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
for nfft in (256, 1024, 1230):
    for avg in (1, 2, 3, 4, 6, 8):
        N = min(1024, nfft)
        rng = np.random.default_rng(nfft)
        freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=N, replace=False)]
        dem = g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=avg, fft_tones=nfft, freq=freq,
                                              wave_type=[g.w_type.TONES] * N), device_index=0)
        out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev)
        for k in range(30):
            dem.process_device(x[k % 4], out)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 300
        for k in range(n):
            dem.process_device(x[k % 4], out)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        print("TONES nfft %5d pf_average %d: %7.2f us per buffer  kernel %s" % (nfft, avg, us, dem.kernel_name), flush=True)
        dem.close()
