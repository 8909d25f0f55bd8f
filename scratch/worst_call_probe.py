"""Which call of the host-pointer submit()/wait() loop of bench.py is the slow one, and is it the
library or the interpreter?  Prints the ten slowest calls (index, ms, split submit / wait) with the
garbage collector on and off."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import gpu_sdr_amd as g
from gpu_sdr_amd.source import host_tones, tone_comb

wl = bench.WORKLOADS["c3"]
rate, N, L = bench.RATE, wl["n_tones"], bench.L
freq, ampl, phase = tone_comb(N, rate, 1)
p = g.param(mode="RX", rate=rate, buffer_len=L, decim=wl["decim"], pf_average=wl["pf_average"],
            freq=[int(f) for f in freq], wave_type=[g.w_type.DIRECT] * N)
for use_gc in (True, False):
    dem = g.RX_buffer_demodulator(p, device_index=0)
    xin = [torch.from_numpy(host_tones(L, i * L, rate, freq[:16], ampl[:16], phase[:16], sigma=1e-3, seed=i)).pin_memory() for i in range(2)]
    outs = [torch.empty(dem.out_capacity, dtype=torch.complex64).pin_memory() for _ in range(3)]
    xn, on = [t.numpy() for t in xin], [t.numpy() for t in outs]
    dem.prepare()
    for _ in range(3):
        dem.process(xn[0], on[0])
    (gc.enable if use_gc else gc.disable)()
    rec, pending = [], 0
    for n in range(1500):
        t0 = time.perf_counter()
        if pending == 3:
            dem.wait(); pending -= 1
        t1 = time.perf_counter()
        dem.submit(xn[n % 2], on[n % 3]); pending += 1
        t2 = time.perf_counter()
        rec.append((n, (t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    while pending:
        dem.wait(); pending -= 1
    rec.sort(key=lambda r: -r[1])
    print("gc", use_gc, "slowest calls (index, total ms, wait ms, submit ms):", [(r[0], round(r[1], 3), round(r[2], 3), round(r[3], 3)) for r in rec[:8]])
    print("   median ms", round(sorted(r[1] for r in rec)[len(rec) // 2], 3))
    dem.close()
