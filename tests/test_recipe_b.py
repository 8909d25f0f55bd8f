"""The C oracle against the second, independent restatement (oracle/recipe_b.py).

Nothing the reference holds pins the oracle ("parity unpinned");
what CAN exist is agreement between two restatements written separately from the
reference source: oracle/gsdr_oracle.c (closed forms over a global sample counter, fp64
accumulate) and oracle/recipe_b.py (vectorised numpy mirroring the reference's buffers,
cuBLAS calls and carry moves).  Tolerance 1e-6 per tone (relative l2), lengths exact.
"""
import numpy as np
import pytest

import oracle
from oracle import recipe_b as rb
from gpu_sdr_amd.source import host_tones, tone_comb


def rel_per_tone(y, yr):
    y = np.asarray(y, dtype=np.complex128)
    yr = np.asarray(yr, dtype=np.complex128)
    return np.linalg.norm(y - yr, axis=0) / np.linalg.norm(yr, axis=0)


def test_sinc_window_two_restatements():
    for length, fc in [(40, 0.0375), (400, 0.75 / 200), (4000, 0.75 / 2000), (1230 * 4, 1 / 2460.), (7, 0.1)]:
        a = oracle.make_sinc_window(length, fc)
        b = rb.make_sinc_window(length, fc)
        assert a.shape == b.shape
        # float sin/cos of numpy and glibc may differ in the last place; both sum to 1
        assert np.max(np.abs(a - b)) <= 4e-7 * np.max(np.abs(a))
        assert abs(float(b.astype(np.float64).sum()) - 1.0) < 1e-5
    # the survey's probe of the reference
    b = rb.make_sinc_window(40, 0.0375)
    assert abs(b[0] - (-0.0013381)) < 2e-7 and abs(b[39] - (-0.0013074)) < 2e-7
    assert abs(b[19] - 0.0768949) < 2e-7 and abs(b[20] - 0.0761853) < 2e-7


def test_flat_window_two_restatements():
    for length, side in [(20, 2), (200, 20), (300, 30), (7, 0), (10, 1)]:
        assert np.array_equal(oracle.make_flat_window(length, side), rb.make_flat_window(length, side))


def test_nco_mix_two_restatements():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    freq = [1000, -25000, 333333, 0, -499999]
    for rate, idx in [(1_000_000, 123456), (1_000_000, 999_000), (200_000_000, 199_999_000)]:
        a = oracle.direct_mix(freq, rate, idx, x)
        b = rb.nco_mix(x, freq, [0] * len(freq), rate, idx)
        assert np.max(np.abs(a - b)) <= 2.5e-7     # float outputs of double arithmetic: <= 1 ulp apart


@pytest.mark.parametrize("acc", [np.complex64, np.complex128])
def test_config1_in_full(acc):
    """BASELINE.json configs[0]: 16 tones, 1 Msample buffers at 100 Msps, decim 100, 4 buffers
    (FIR carry and the NCO index wrap at the sample rate are both exercised)."""
    rate, L, M, F, N = 100_000_000, 1_000_000, 100, 4, 16
    freq, ampl, phase = tone_comb(N, rate, seed=20251004)
    ref = oracle.Direct(freq, rate, M, F, L)
    alt = rb.Direct(freq, rate, M, F, L, acc=acc)
    assert np.max(np.abs(ref.taps() - alt.taps)) <= 4e-7 * np.max(np.abs(ref.taps()))
    worst = 0.0
    for c in range(4):
        x = host_tones(L, c * L, rate, freq, ampl, phase, sigma=1e-3, seed=100 + c)
        ya, yb = ref.process(x), alt.process(x)
        assert ya.shape == yb.shape == (L // M, N)
        worst = max(worst, float(rel_per_tone(yb, ya).max()))
    assert worst <= 1e-6, worst
    # and the closed form: a pure tone at f_k demodulates to a_k*exp(i*phase_k) (sum of taps = 1)
    want = ampl * np.exp(1j * phase)
    got = yb[F:].mean(axis=0)
    assert np.max(np.abs(got - want)) < 2e-3 * np.max(np.abs(want))


def test_direct_undecimated_and_odd_shapes():
    for (rate, L, M, F, N) in [(1_000_000, 50_000, 0, 4, 3), (1_000_000, 50_000, 50, 1, 5), (200_000_000, 60_000, 1000, 4, 9)]:
        freq, ampl, phase = tone_comb(N, rate, seed=5)
        ref = oracle.Direct(freq, rate, M, F, L)
        alt = rb.Direct(freq, rate, M, F, L, acc=np.complex128)
        for c in range(3):
            x = host_tones(L, c * L, rate, freq, ampl, phase, sigma=1e-3, seed=7 + c)
            ya, yb = ref.process(x), alt.process(x)
            assert ya.shape == yb.shape
            assert float(rel_per_tone(yb, ya).max()) <= 1e-6


@pytest.mark.parametrize("nfft,avg,L,N", [(10, 4, 103, 3), (100, 4, 50_000, 8), (1230, 4, 200_000, 32), (1000, 1, 50_000, 5)])
def test_pfb_two_restatements(nfft, avg, L, N):
    """TONES through numpy.fft, with the reference's raw_input / move_buffer mechanics:
    lengths per call must be exact (the client trusts them), values <= 1e-6."""
    rate = 1_000_000
    rng = np.random.default_rng(11)
    bins = sorted(rng.choice(nfft, size=min(N, nfft), replace=False).tolist())
    freq = [int((b if b < nfft // 2 else b - nfft) * (rate // nfft)) for b in bins]
    ref = oracle.Pfb(freq, rate, nfft, avg, L)
    alt = rb.Pfb(freq, rate, nfft, avg, L)
    assert list(ref.bins()) == alt.bins
    assert ref.batching == alt.batching
    for c in range(5):
        x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
        ya, yb = ref.process(x), alt.process(x)
        assert ya.shape == yb.shape, (c, ya.shape, yb.shape)
        if ya.size:
            assert float(rel_per_tone(yb, ya).max()) <= 1e-6


def test_pfb_survey_probe_lengths():
    """nfft=10, avg=4, L=103 -> current_batch 7,10,10,11,10; new_0 0,33,36,39,32."""
    alt = rb.Pfb([0], 1000, 10, 4, 103, bins=[0])
    cb, n0 = [], []
    for _ in range(5):
        n0.append(alt.buf.new_0)
        cb.append(alt.process(np.zeros(103, np.complex64)).shape[0])
    assert cb == [7, 10, 10, 11, 10] and n0 == [0, 33, 36, 39, 32]


def test_tone_bins_two_restatements():
    rng = np.random.default_rng(2)
    for rate, nfft in [(1_000_000, 100), (200_000_000, 1230), (100_000_000, 7)]:
        freq = [int(v) for v in rng.integers(-rate // 2 + 1, rate // 2, size=40)] + [0, rate // nfft, -(rate // nfft)]
        assert list(oracle.pfb_tone_bins(rate, nfft, freq)) == rb.pfb_tone_bins(rate, nfft, freq)


@pytest.mark.parametrize("args", [
    (200_000_000, -100_000_000, 100_000_000, 1_000_000, 1.0),      # C4
    (200_000_000, -100_000_000, 100_000_000, 1_000_000, 1.5),      # carry variant, length 300
    (200_000_000, -90_000_000, 90_000_000, 1000, 3.5e-5),          # survey probe: length 7
    (100_000_000, 10_000_000, -40_000_000, 0, 1e-3),               # swipe_s unset, downward chirp
])
def test_chirp_params_two_restatements(args):
    cp = oracle.chirp_params(*args)
    assert (cp.num_steps, cp.length, cp.chirpness, cp.f0) == rb.chirp_params(*args)


def test_chirp_survey_probe():
    assert rb.chirp_params(200_000_000, -90_000_000, 90_000_000, 1000, 3.5e-5) == (1000, 7, 3869339, -1932735282)


@pytest.mark.parametrize("decim,chirp_t,L", [(0, 1.0, 50_000), (1, 1.0, 100_000), (1, 1.5, 100_000), (2, 1.0, 60_000)])
def test_chirp_two_restatements(decim, chirp_t, L):
    rate = 200_000_000
    rng = np.random.default_rng(9)
    ref = oracle.Chirp(rate, -rate // 2, rate // 2, 1_000_000, chirp_t, decim, L)
    alt = rb.Chirp(rate, -rate // 2, rate // 2, 1_000_000, chirp_t, decim, L, acc=np.complex128)
    cp = oracle.chirp_params(rate, -rate // 2, rate // 2, 1_000_000, chirp_t)
    for c in range(4):
        x = (oracle.chirp_gen(cp, c * L, L, 0.5)
             + 1e-3 * (rng.standard_normal(L) + 1j * rng.standard_normal(L))).astype(np.complex64)
        ya, yb = ref.process(x), alt.process(x)
        assert ya.shape == yb.shape, (c, ya.shape, yb.shape)
        err = np.linalg.norm(ya.astype(np.complex128) - yb) / np.linalg.norm(ya.astype(np.complex128))
        assert err <= 1e-6, (c, err)


def test_chirp_index_wrap_large_period():
    """period > 2^32 and indices near the 64-bit wrap of the phase product."""
    ns, ln, ch, f0 = 3_000_000, 2000, 4294, -2147483647
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64)
    cp = oracle.ChirpParam(ns, ln, ch, f0)
    for last in [0, 5_999_990_000, 2 ** 32 - 100, 4_500_000_123]:
        a = oracle.chirp_demod(cp, last, x)
        b = rb.chirp_demod(x, last, ns, ln, ch, f0)
        assert np.max(np.abs(a - b)) <= 1e-6 * np.max(np.abs(a))


# ---------------------------------------------------------------------------
# TX tone comb (row f3): oracle (closed form, exact integer phase) against the literal
# restatement (assignment into a length-`rate` vector + unnormalised inverse FFT) and
# against the host helper the product uses (gsdr_tx_tone_bins)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("rate", [4096, 10_000])
def test_tone_gen_two_restatements(rate, gsdr_lib):
    from gpu_sdr_amd.generator import tone_bins
    from gpu_sdr_amd.source import host_tones
    freq = [100, -250, 0, 777, 100, -rate, rate, rate // 2, -250, -1]     # 0 Hz, duplicates, +-rate, Nyquist
    ampl = [0.1, 0.2, 0.3, 0.05, 0.4, 0.07, 0.9, 0.11, 0.6, 0.02]
    lit = rb.tone_gen(freq, ampl, rate)                                   # the whole periodic buffer
    assert lit.shape == (rate,)
    for start, n in [(0, rate), (rate - 100, 300), (5 * rate + 7, 1000)]:
        a = oracle.tone_gen(freq, ampl, rate, start, n)
        b = lit[(start + np.arange(n)) % rate]
        assert np.max(np.abs(a - b)) <= 2e-6, (start, n)
    # the quirks, spelled out: the 0 Hz tone (0.3) and f = +rate (0.9) are not generated, f = -rate
    # IS the DC term (0.07); of the duplicates the last amplitude wins (0.4, 0.6), they do not add
    spec = np.fft.fft(lit.astype(np.complex128)) / rate
    want = {100: 0.4, rate - 250: 0.6, 777: 0.05, 0: 0.07, rate // 2: 0.11, rate - 1: 0.02}
    for b_, a_ in want.items():
        assert abs(spec[b_] - a_) < 1e-6, b_
    others = np.ones(rate, bool)
    others[list(want)] = False
    assert np.max(np.abs(spec[others])) < 1e-6
    # the product's host helper agrees tone for tone
    f2, a2 = tone_bins(freq, ampl, rate)
    got = {int(f) % rate: float(a) for f, a in zip(f2, a2)}
    assert got.keys() == want.keys() and all(abs(got[k] - want[k]) < 1e-7 for k in want)
    x = host_tones(300, rate - 100, rate, f2, a2, np.zeros(len(f2), np.float32))
    assert np.max(np.abs(x - oracle.tone_gen(freq, ampl, rate, rate - 100, 300))) <= 2e-6
