"""NOISE through the in-LDS kernels against numpy's FFT of the same polyphase-filtered frames, per bin."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g
dev = torch.device("cuda:0")
for nfft, avg, L in [(17, 3, 700), (17, 1, 5000), (333, 2, 4000), (34, 3, 3000), (101, 2, 5000), (1230, 4, 100000), (1016, 2, 40000), (37 * 8, 4, 30000)]:
    rng = np.random.default_rng(nfft)
    p = g.param(mode="RX", rate=1_000_000, buffer_len=L, decim=0, pf_average=avg, fft_tones=nfft, freq=[0], wave_type=[g.w_type.NOISE])
    dem = g.RX_buffer_demodulator(p, device_index=0)
    w = dem.window().astype(np.float64)
    x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
    out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev)
    n = dem.process_device(torch.from_numpy(x).to(dev), out)
    torch.cuda.synchronize()
    y = out[:n].cpu().numpy().reshape(-1, nfft)
    fr = y.shape[0]
    frames = np.stack([sum(x[(r + i) * nfft:(r + i + 1) * nfft].astype(np.complex128) * w[i * nfft:(i + 1) * nfft] for i in range(avg)) for r in range(fr)])
    ref = np.fft.fft(frames, axis=1)
    err = np.abs(y - ref).max(axis=0) / np.abs(ref).max()
    bad = np.nonzero(err > 1e-4)[0]
    print(nfft, avg, L, dem.kernel_name, "frames", fr, "worst", float(err.max()), "bad bins", bad[:20].tolist(), len(bad))
    dem.close()
