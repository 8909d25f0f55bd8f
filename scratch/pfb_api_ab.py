"""TONES per 1 M-sample buffer: the in-order entry (process_device) against the overlapped one
(submit_device / wait, three buffers outstanding), device-resident."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g

dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
for nfft in [int(v) for v in sys.argv[1:]] or [256, 1024, 1230, 2048]:
    N = min(1024, nfft)
    rng = np.random.default_rng(nfft)
    freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=N, replace=False)]
    for api in ("inorder", "overlapped"):
        p = g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=4, fft_tones=nfft, freq=freq, wave_type=[g.w_type.TONES] * N)
        dem = g.RX_buffer_demodulator(p, device_index=0)
        dem.prepare(host=False, pipeline=True, pipeline_host=False, rehearse=False)
        outs = [torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev) for _ in range(4)]
        def run(n):
            if api == "inorder":
                for k in range(n):
                    dem.process_device(x[k % 4], outs[k % 4])
            else:
                pending = 0
                for k in range(n):
                    dem.submit_device(x[k % 4], outs[k % 4])
                    pending += 1
                    if pending == 3:
                        dem.wait(); pending -= 1
                while pending:
                    dem.wait(); pending -= 1
            torch.cuda.synchronize()
        run(60)
        t0 = time.perf_counter()
        run(600)
        us = (time.perf_counter() - t0) / 600 * 1e6
        print("TONES nfft %5d %-10s %7.2f us per buffer  kernel %s" % (nfft, api, us, dem.kernel_name), flush=True)
        dem.close()
