"""Two front-ends on one GPU: a heavy matrix-core DIRECT handle streams on one thread while handles of the
other modes (chirp lock-in, TONES through the in-LDS FFT, a packed-FP32 DDC) run on their own threads and
streams; every result of the small handles is compared with a reference result computed while the GPU was
otherwise idle.  Looks for cross-kernel hazards (rule R3: packed FP32 beside an MFMA loop on one SIMD)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import gpu_sdr_amd as g

dev = torch.device("cuda:0")
rate = 200_000_000
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0


def mk_chirp(L):
    return g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=1, freq=[-rate // 2], chirp_f=[rate // 2],
                                           swipe_s=[1_000_000], chirp_t=[1.0], wave_type=[g.w_type.CHIRP]), device_index=0)


def mk_tones(L):
    rng = np.random.default_rng(5)
    freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=256, replace=False)]
    return g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=4, fft_tones=1230, freq=freq,
                                           wave_type=[g.w_type.TONES] * 256), device_index=0)


def mk_flat(L):
    os.environ["GSDR_DDC_MFMA"] = "0"
    try:
        freq = [1_000_000 * (k + 1) for k in range(64)]
        return g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=100, pf_average=4, freq=freq,
                                               wave_type=[g.w_type.DIRECT] * 64), device_index=0)
    finally:
        del os.environ["GSDR_DDC_MFMA"]


def mk_mix(L):
    freq = [1_000_000 * (k + 1) for k in range(8)]
    return g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=1, freq=freq,
                                           wave_type=[g.w_type.DIRECT] * 8), device_index=0)


def mk_direct(L):
    rng = np.random.default_rng(11)
    freq = [int(f) for f in rng.choice(np.arange(-rate // 2 + 1, rate // 2), size=256, replace=False)]
    return g.RX_buffer_demodulator(g.param(mode="RX", rate=rate, buffer_len=L, decim=100, pf_average=4, freq=freq,
                                           wave_type=[g.w_type.DIRECT] * 256), device_index=0)


L = 200_000
cases = {"chirp": mk_chirp, "tones": mk_tones, "flat": mk_flat, "mix": mk_mix, "direct": mk_direct}
xs = [(torch.randn(L, device=dev) + 1j * torch.randn(L, device=dev)).to(torch.complex64) for _ in range(4)]
NB = 12
# reference pass: GPU otherwise idle; the stateful modes run the same NB buffers from a fresh handle every round
refs = {}
for name, mk in cases.items():
    dem = mk(L)
    out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev)
    r = []
    for k in range(NB):
        n = dem.process_device(xs[k % 4], out)
        torch.cuda.synchronize()
        r.append(out[:n].clone())
    refs[name] = r
    dem.close()

stop = threading.Event()
create_lock = threading.Lock()      # mk_flat() changes the environment: one creation at a time
bad = {k: 0 for k in cases}
rounds = {k: 0 for k in cases}


def explain_chirp(k, v, got, want):
    """which samples of point v would explain got - want?  (per-sample contributions from the numpy restatement)"""
    from oracle import recipe_b
    ns, length, chirpness, f0 = recipe_b.chirp_params(rate, -rate // 2, rate // 2, 1_000_000, 1.0)
    ppt = length
    x = xs[k % 4].cpu().numpy()
    dm = recipe_b.chirp_demod(x, (k * L) % (ns * length), ns, length, chirpness, f0)
    w = recipe_b.make_flat_window(ppt, ppt // 10)
    c = dm[v * ppt:(v + 1) * ppt].astype(np.complex128) * w
    d = got - want
    print("   point", v, "got", got, "want", want, "sum of contributions", c.sum(), "diff", d)
    best = []
    for g0 in range(0, ppt, 16):
        for span in (16, 32, 64):
            sub = c[g0:g0 + span].sum()
            for sign in (1, -1):
                best.append((abs(d - sign * sub), sign, g0, span))
    for s0 in range(ppt):
        for sign in (1, -1, 2, -2):
            best.append((abs(d - sign * c[s0]), sign, s0, 1))
    best.sort()
    print("   best explanations (residual, sign, first sample, span):", [(round(b[0], 6), b[1], b[2], b[3]) for b in best[:4]], flush=True)


def heavy():
    eng = bench.HipEngine()
    dem, bufs, outs, N = eng.build(bench.WORKLOADS["c3"], dev, 7)
    k, pend = 0, 0
    while not stop.is_set():
        if pend == 3:
            dem.wait(); pend -= 1
        dem.submit_device(bufs[k % 8], outs[k % 3]); pend += 1; k += 1
    while pend:
        dem.wait(); pend -= 1
    dem.close()
    print("heavy: %d buffers of C3" % k, flush=True)


def small(name):
    # everything of this thread -- allocations, comparisons, the library's launches -- on ONE stream of its
    # own: torch's caching allocator hands a freed block to the next taker on the stream it was allocated
    # on, so a buffer allocated on the shared default stream and written through a side stream can be
    # overwritten by another thread's pending default-stream kernel (it was: 0x01 bytes of a comparison
    # result in the chirp output)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        while not stop.is_set():
            with create_lock:
                dem = cases[name](L)
            out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=dev)
            for k in range(NB):
                n = dem.process_device(xs[k % 4], out, st)
                st.synchronize()
                if not torch.equal(out[:n], refs[name][k]):
                    bad[name] += 1
                    d = (out[:n] - refs[name][k]).abs()
                    print(name, "MISMATCH round", rounds[name], "buffer", k, "max abs diff", float(d.max()), "at", int(d.argmax()),
                          "differing elements", int((d > 0).sum()), flush=True)
                    if name == "chirp":
                        explain_chirp(k, int(d.argmax()), complex(out[int(d.argmax())].item()), complex(refs[name][k][int(d.argmax())].item()))
            dem.close()
            rounds[name] += 1


th = [threading.Thread(target=heavy)] + [threading.Thread(target=small, args=(n,)) for n in cases]
for t in th:
    t.start()
time.sleep(SECONDS)
stop.set()
for t in th:
    t.join()
print("rounds", rounds, "mismatching buffers", bad, flush=True)
sys.exit(1 if any(bad.values()) else 0)
