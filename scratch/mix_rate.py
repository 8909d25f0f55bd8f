"""DIRECT without decimation (the NCO mix alone, ref cpp/kernels.cu:45-86): HBM write bound, 8 B per tone-sample."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g
dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
for N in (1, 2, 4, 8, 16, 32, 64):
    rng = np.random.default_rng(N)
    freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=N, replace=False)]
    dem = g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=4, freq=freq, wave_type=[g.w_type.DIRECT] * N), device_index=0)
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
    by = 8.0 * L * (N + 1)
    print("mix N=%3d: %8.2f us per buffer, %7.1f GB/s algorithmic (write %d MB), kernel %s" % (N, us, by / us / 1e3, 8 * L * N // 1000000, dem.kernel_name), flush=True)
    dem.close()
