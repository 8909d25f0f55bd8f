"""Chirp VNA lock-in over points per frequency step (ppt = chirp_t * rate / swipe_s): microseconds per 1 M-sample buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sdr_amd as g
dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
for swipe in (20_000_000, 4_000_000, 1_000_000, 100_000, 20_000, 2_000, 200):
    p = g.param(mode="RX", rate=rate, buffer_len=L, decim=1, freq=[-rate // 2], chirp_f=[rate // 2], swipe_s=[swipe], chirp_t=[1.0],
                wave_type=[g.w_type.CHIRP])
    try:
        dem = g.RX_buffer_demodulator(p, device_index=0)
    except Exception as e:
        print("chirp swipe_s %9d: refused (%s)" % (swipe, e)); continue
    out = torch.empty(max(dem.out_capacity, 1), dtype=torch.complex64, device=dev)
    for k in range(20):
        dem.process_device(x[k % 4], out)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for k in range(n):
        dem.process_device(x[k % 4], out)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print("chirp swipe_s %9d (ppt %7d): %8.2f us per buffer  %7.1f GB/s  kernel %s" % (swipe, rate // swipe, us, 8.0 * L / us / 1e3, dem.kernel_name), flush=True)
    dem.close()
