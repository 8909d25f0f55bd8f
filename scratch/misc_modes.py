"""The small modes on 1 Mi-sample buffers, in order on one stream: undecimated chirp demodulation
(8 B read + 8 B written per sample), undecimated DIRECT mix (8 B read, 8 N B written), NODSP copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sdr_amd as g

dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
cases = [
    ("chirp, no lock-in (decim 0)", g.param(mode="RX", rate=rate, buffer_len=L, decim=0, freq=[-rate // 2], chirp_f=[rate // 2],
                                            swipe_s=[1_000_000], chirp_t=[1.0], wave_type=[g.w_type.CHIRP]), 16.0),
    ("DIRECT mix, 8 tones (decim 0)", g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=1,
                                              freq=[1_000_000 * (k + 1) for k in range(8)], wave_type=[g.w_type.DIRECT] * 8), 8.0 + 64.0),
    ("NODSP", g.param(mode="RX", rate=rate, buffer_len=L, decim=0, freq=[0], wave_type=[g.w_type.NODSP]), 16.0),
]
for name, p, bytes_per_sample in cases:
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
    print("%-32s %7.2f us per buffer  %8.0f Msamples/s  %.2f TB/s (%.0f B per sample)  kernel %s" % (
        name, us, L / us, bytes_per_sample * L / us / 1e6, bytes_per_sample, dem.kernel_name), flush=True)
    dem.close()
