"""TONES (and NOISE) through the frame-per-workgroup kernel: microseconds per 1 Mi-sample buffer
over frame lengths of different factorisations, in order on one stream."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g

dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
SIZES = [int(v) for v in sys.argv[1:]] or [64, 256, 1000, 1024, 1200, 1230, 1250, 2048, 17 * 64, 41 * 32, 127 * 8, 4096, 8192]
for nfft in SIZES:
    for mode in ("TONES", "NOISE"):
        N = min(1024, nfft)
        rng = np.random.default_rng(nfft)
        freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=N, replace=False)]
        if mode == "TONES":
            p = g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=4, fft_tones=nfft, freq=freq,
                        wave_type=[g.w_type.TONES] * N)
        else:
            p = g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=4, fft_tones=nfft, freq=[0],
                        wave_type=[g.w_type.NOISE])
        dem = g.RX_buffer_demodulator(p, device_index=0)
        out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev)
        for k in range(50):
            dem.process_device(x[k % 4], out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        for k in range(n):
            dem.process_device(x[k % 4], out)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        print("%-5s nfft %5d: %7.2f us per buffer  %8.0f Msamples/s  kernel %s" % (mode, nfft, us, L / us, dem.kernel_name), flush=True)
        dem.close()
