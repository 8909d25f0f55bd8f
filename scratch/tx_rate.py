"""TX tone comb generation rate: TX_buffer_generator(TONES).get() on 1 Mi-sample buffers (gsdr_txgen_*)
against the per-sample synthesis of the bench's input source (gsdr_source_tones)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g
from gpu_sdr_amd.source import device_tones, tone_comb

dev = torch.device("cuda:0")
rate, L = 200_000_000, 1_000_000
x = torch.empty(L, dtype=torch.complex64, device=dev)
for N in (16, 256, 2048, 8192):
    freq, ampl, phase = tone_comb(N, rate, 3)
    tx = g.TX_buffer_generator(g.param(mode="TX", rate=rate, buffer_len=L, freq=[int(f) for f in freq], ampl=[float(a) for a in ampl],
                                       wave_type=[g.w_type.TONES] * N))
    for _ in range(3):
        tx.get(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 50
    for _ in range(n):
        tx.get(x)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    device_tones(x, 0, rate, freq, ampl, phase, sigma=0.0)
    torch.cuda.synchronize()
    us_old = (time.perf_counter() - t0) * 1e6
    print("%5d tones: %8.1f us per 1 Mi-sample buffer = %7.0f Msamples/s (%.1f x real time at 200 Msps); per-sample synthesis %9.1f us" % (
        N, us, L / us, L / us / 200.0, us_old), flush=True)
    tx.close()
