"""DIRECT (decimated) over tone counts and decimations off the bench's workloads: microseconds per 1 M-sample buffer,
in order, and what the algorithmic traffic 8 (1 + N / M) B per sample amounts to."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g
dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
for M in (10, 100, 1000):
    for N in (1, 4, 16, 64, 256, 1024):
        if N * (L // M) > 40_000_000:
            continue
        rng = np.random.default_rng(N + M)
        freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=N, replace=False)]
        dem = g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=M, pf_average=4, freq=freq, wave_type=[g.w_type.DIRECT] * N), device_index=0)
        out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev)
        for k in range(20):
            dem.process_device(x[k % 4], out)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for k in range(n):
            dem.process_device(x[k % 4], out)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        gf = N * (6 + 4 * 4) * L / us / 1e6
        print("DIRECT decim %5d N=%5d: %8.2f us per buffer  %7.1f GB/s algorithmic  %8.1f TFLOP/s algorithmic  kernel %s" % (
            M, N, us, 8.0 * L * (1 + N / M) / us / 1e3, gf, dem.kernel_name), flush=True)
        dem.close()
