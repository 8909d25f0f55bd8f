"""Soak of the in-LDS TONES kernels (run kernel: direct filter, teams with their LDS arrival counters): two handles
with the same parameters take the same buffers, one through process_device on torch's stream, one through
submit_device / wait on the library's -- their launches meet on the chip -- and every output is compared bit for
bit (fingerprint of every buffer, the full buffer every 50th).  A race in a team barrier shows here or nowhere."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sdr_amd as g

dev = torch.device("cuda:0")
L, rate = 1_000_000, 200_000_000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
x = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) * s for s in (1e-3, 1.0, 40.0, 1.0, 1e-2, 5.0, 1.0, 0.3)]
total_bad = 0
for nfft in (256, 1000, 1024, 1230, 2048, 2560, 4096):
    N = min(1024, nfft)
    rng = np.random.default_rng(nfft)
    freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=N, replace=False)]
    mk = lambda: g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=4, fft_tones=nfft, freq=freq,
                                                 wave_type=[g.w_type.TONES] * N), device_index=0)
    da, db = mk(), mk()
    db.prepare(host=False, pipeline=True, pipeline_host=False, rehearse=False)
    out_a = torch.empty(da.out_capacity, dtype=torch.complex64, device=dev)
    outs = [torch.empty(db.out_capacity, dtype=torch.complex64, device=dev) for _ in range(3)]
    bad, pend, t0 = 0, [], time.time()
    def check():
        global bad
        kk, fp, fl = pend.pop(0)
        m = db.wait()
        o = outs[kk % 3][:m]
        got = torch.view_as_real(o).view(torch.int32).sum(dtype=torch.int64)
        if got.item() != fp.item() or (fl is not None and not torch.equal(o, fl)):
            bad += 1
            print(nfft, "MISMATCH at buffer", kk, flush=True)
    for k in range(steps):
        db.submit_device(x[k % 8], outs[k % 3])                 # on the library's stream ...
        n = da.process_device(x[k % 8], out_a)                  # ... while this one runs on torch's
        full = out_a[:n].clone() if k % 50 == 0 else None
        fp = torch.view_as_real(out_a[:n]).view(torch.int32).sum(dtype=torch.int64)
        pend.append((k, fp, full))
        if len(pend) == 3:
            check()
    while pend:
        check()
    torch.cuda.synchronize()
    print("nfft %5d: %d buffers x 2 handles, %d mismatches, %.1f s, kernel %s" % (nfft, steps, bad, time.time() - t0, da.kernel_name), flush=True)
    total_bad += bad
    da.close(); db.close()
sys.exit(1 if total_bad else 0)
