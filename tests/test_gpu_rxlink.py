"""The C++ drop-in class, executed: tools/rx_link.cpp is the reference's rx_single_link loop
(ref: cpp/USRP_server_link_threads.cpp:647-690) around include/USRP_demodulator.hpp -- the
header a GPU_SDR server tree would compile against -- built by gpu_sdr_amd/csrc/Makefile.
Here it runs over a recorded, seeded stream and its packet payloads and lengths are compared
with the CPU oracle: <= 1e-5 per tone, lengths exact."""
import json
import os
import subprocess

import numpy as np
import pytest

from _margins import record_margin

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RX_LINK = os.path.join(ROOT, "gpu_sdr_amd", "rx_link")
TOL = 1e-5


def run_rx_link(tmp_path, cfg_lines, x, pipe):
    assert os.path.exists(RX_LINK), "gpu_sdr_amd/rx_link is built by __graft_entry__.build() / make -C gpu_sdr_amd/csrc"
    cfg, fin, fout = tmp_path / "cfg.txt", tmp_path / "in.c64", tmp_path / "out.c64"
    cfg.write_text("\n".join(cfg_lines) + "\n")
    np.ascontiguousarray(x, dtype=np.complex64).tofile(fin)
    cmd = [RX_LINK, "file", str(cfg), str(fin), str(fout)] + (["pipe"] if pipe else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    info = json.loads(p.stdout.strip().splitlines()[-1])
    return info, np.fromfile(fout, dtype=np.complex64)


@pytest.mark.parametrize("pipe", [False, True], ids=["process", "submit_wait"])
def test_rx_link_direct_against_oracle(cuda_device, gsdr_lib, oracle_mod, tmp_path, pipe):
    """DIRECT through `new RX_buffer_demodulator(&param)` + process()/submit()+wait():
    7 buffers (more than the pipeline depth), FIR carry and NCO index across all of them."""
    from gpu_sdr_amd.source import host_tones, tone_comb
    N, rate, M, F, L, nbuf = 48, 10_000_000, 100, 4, 100_000, 7
    freq, ampl, phase = tone_comb(N, rate, seed=4711)
    x = np.concatenate([host_tones(L, c * L, rate, freq, ampl, phase, sigma=1e-3, seed=c) for c in range(nbuf)])
    info, y = run_rx_link(tmp_path, ["mode DIRECT", f"rate {rate}", f"buffer_len {L}", f"decim {M}",
                                     f"pf_average {F}", "freq " + " ".join(str(int(f)) for f in freq)], x, pipe)
    ref = oracle_mod.Direct(freq, rate, M, F, L)
    yr = np.concatenate([ref.process(x[c * L:(c + 1) * L]) for c in range(nbuf)])
    assert info["channels"] == N and info["lengths"] == [N * (L // M)] * nbuf
    assert y.size == yr.size
    err = np.linalg.norm(y.reshape(-1, N) - yr, axis=0) / np.linalg.norm(yr, axis=0)
    record_margin(float(err.max()))
    assert err.max() <= TOL


@pytest.mark.parametrize("pipe", [False, True], ids=["process", "submit_wait"])
@pytest.mark.parametrize("chirp_t,decim", [(1.0, 1), (1.5, 1), (1.0, 0)], ids=["ppt200", "ppt300_carry", "undecimated"])
def test_rx_link_chirp_against_oracle(cuda_device, gsdr_lib, oracle_mod, tmp_path, pipe, chirp_t, decim):
    """CHIRP (VNA) through the class: lock-in with and without a carried remainder
    (ppt 300 does not divide the buffer), and the undecimated demodulator."""
    rate, L, nbuf, steps = 200_000_000, 100_000, 6, 1_000_000
    cp = oracle_mod.chirp_params(rate, -rate // 2, rate // 2, steps, chirp_t)
    rng = np.random.default_rng(5)
    x = np.concatenate([oracle_mod.chirp_gen(cp, c * L, L, 0.5) for c in range(nbuf)])
    x = (x + 1e-3 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    info, y = run_rx_link(tmp_path, ["mode CHIRP", f"rate {rate}", f"buffer_len {L}", f"decim {decim}",
                                     f"freq {-rate // 2}", f"chirp_f {rate // 2}", f"swipe_s {steps}",
                                     f"chirp_t {chirp_t}"], x, pipe)
    ref = oracle_mod.Chirp(rate, -rate // 2, rate // 2, steps, chirp_t, decim, L)
    outs = [ref.process(x[c * L:(c + 1) * L]) for c in range(nbuf)]
    assert info["lengths"] == [len(o) for o in outs]
    yr = np.concatenate(outs)
    assert y.size == yr.size
    err = float(np.linalg.norm(y - yr) / np.linalg.norm(yr))
    record_margin(err)
    assert err <= TOL


def test_rx_link_tones_against_oracle(cuda_device, gsdr_lib, oracle_mod, tmp_path):
    """TONES (PFB) through the class: the valid length changes from packet to packet."""
    rate, nfft, avg, L, nbuf = 1_000_000, 100, 4, 100_037, 6
    bins = [0, 3, 17, 50, 77, 99]
    freq = [int((b if b < nfft // 2 else b - nfft) * (rate // nfft)) for b in bins]
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(L * nbuf) + 1j * rng.standard_normal(L * nbuf)).astype(np.complex64)
    info, y = run_rx_link(tmp_path, ["mode TONES", f"rate {rate}", f"buffer_len {L}", "decim 0", f"pf_average {avg}",
                                     f"fft_tones {nfft}", "freq " + " ".join(map(str, freq))], x, True)
    ref = oracle_mod.Pfb(freq, rate, nfft, avg, L)
    outs = [ref.process(x[c * L:(c + 1) * L]) for c in range(nbuf)]
    assert info["lengths"] == [o.size for o in outs]
    assert len(set(info["lengths"])) > 1
    yr = np.concatenate([o.reshape(-1) for o in outs]).reshape(-1, len(freq))
    err = np.linalg.norm(y.reshape(-1, len(freq)) - yr, axis=0) / np.linalg.norm(yr, axis=0)
    record_margin(float(err.max()))
    assert err.max() <= TOL


def test_rx_link_first_calls_are_not_late(cuda_device, gsdr_lib):
    """The throughput harness (synthetic RX thread, pinned pools, streamer stand-in): with
    everything created in the constructor and the process-wide first-use costs paid there by a
    rehearsal (gsdr_demod_prepare with GSDR_PREPARE_REHEARSE), no call of the pipelined loop may take
    a buffer period -- round 1's first calls took ten (47-49 ms), round 2's before the rehearsal
    one and a half (5 - 7 ms on calls 0, 1 and 6)."""
    p = subprocess.run([RX_LINK, "256", "100", "300", "pipe"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    info = json.loads(p.stdout.strip().splitlines()[-1])
    assert info["streamed_samples"] == 300 * (1_000_000 // 100) * 256
    # 1 M samples at 200 Msps = 5 ms per buffer.  Measured: 0.55 - 1.0 ms for the slowest call of a run
    # (profiles/r02_rxlink.log).  Wall-clock bounds on a shared box can fail without any code change, so what
    # gates is only the gross regression (round 1: ten buffer periods); the figures themselves go to
    # gpurun_out/ and the tighter expectations below are reported as warnings
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rxlink_latency.json"), "w") as f:
        json.dump(info, f)
    assert info["worst_ms"] < 40.0, info
    assert info["worst_ms_after_first_8_calls"] < 25.0, info
    if info["worst_ms"] >= 10.0 or info["calls_above_3ms"] > 1 or info["worst_ms_after_first_8_calls"] >= 5.0:
        import warnings
        warnings.warn(f"rx_link: slower calls than usual (expected worst < 10 ms, at most one call above 3 ms, "
                      f"< 5 ms after the first 8): {info}")


def run_tx(tmp_path, cfg_lines, nbuf):
    cfg, fout = tmp_path / "txcfg.txt", tmp_path / "tx.c64"
    cfg.write_text("\n".join(cfg_lines) + "\n")
    p = subprocess.run([RX_LINK, "tx", str(cfg), str(fout), str(nbuf)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), np.fromfile(fout, dtype=np.complex64)


def test_tx_class_tones_and_chirp_against_oracle(cuda_device, gsdr_lib, oracle_mod, tmp_path):
    """`new TX_buffer_generator(&param)` + get() from include/USRP_buffer_generator.hpp (the reference's TX
    class, over gsdr_txgen_*), driven by the reference's tx_single_link loop (tools/rx_link.cpp, tx mode;
    ref: cpp/USRP_server_link_threads.cpp:568-584): memory per packet only when param::dynamic_buffer() says so.
    For TONES that loop hands get() an UNALLOCATED pointer (poisoned in the harness: writing through it faults)
    and takes back a pointer into the generator's own period buffer (ref: get_from_tones,
    cpp/USRP_buffer_generator.cpp:226-229); for CHIRP get() fills the caller's buffer.  Buffer by buffer: the
    tone comb against the oracle's tone_gen (bin assignment quirks included, buffers wrapping the period), the
    chirp against its chirp_gen with the TX side's own parameter derivation -- including a sweep whose steps
    would be shorter than a sample, where the TX side resets num_steps before it takes the slope (:118-125)."""
    rate, L, nbuf = 1_000_000, 300_007, 5
    freq = [1000, -250_000, 0, 77_777, 1000, -rate, rate, rate // 2, -5, -1]
    ampl = [0.1, 0.2, 0.3, 0.05, 0.4, 0.07, 0.9, 0.11, 0.6, 0.02]
    info, y = run_tx(tmp_path, ["mode TONES", f"rate {rate}", f"buffer_len {L}", "freq " + " ".join(map(str, freq)),
                                "ampl " + " ".join(map(str, ampl))], nbuf)
    assert info["buffers"] == nbuf and y.size == nbuf * L
    start = 0
    for c in range(nbuf):
        want = oracle_mod.tone_gen(freq, ampl, rate, start, L)
        err = float(np.max(np.abs(y[c * L:(c + 1) * L] - want)))
        record_margin(err / float(np.sum(ampl)), "tones: max abs error / sum of amplitudes")
        assert err <= 2e-6 * float(np.sum(ampl)), (c, err)
        start = (start + L) % rate
    rate, L, nbuf, steps, t = 200_000_000, 100_000, 4, 10_000, 0.0075
    info, y = run_tx(tmp_path, ["mode CHIRP", f"rate {rate}", f"buffer_len {L}", "freq -80000000", "chirp_f 80000000",
                                f"swipe_s {steps}", f"chirp_t {t}", "ampl 0.25"], nbuf)
    cp = oracle_mod.chirp_params_tx(rate, -80_000_000, 80_000_000, steps, t)
    last = 0
    for c in range(nbuf):
        want = oracle_mod.chirp_gen(cp, last, L, 0.25)
        np.testing.assert_allclose(y[c * L:(c + 1) * L], want, rtol=0, atol=3e-7)
        last = (last + L) % (cp.num_steps * cp.length)
    # steps shorter than a sample: chirp_t * rate = 4000 < swipe_s = 10000 -> length 1, num_steps 4000, slope from 4000
    t = 2e-5
    info, y = run_tx(tmp_path, ["mode CHIRP", f"rate {rate}", f"buffer_len {L}", "freq -80000000", "chirp_f 80000000",
                                f"swipe_s {steps}", f"chirp_t {t}", "ampl 0.25"], nbuf)
    cp = oracle_mod.chirp_params_tx(rate, -80_000_000, 80_000_000, steps, t)
    rx = oracle_mod.chirp_params(rate, -80_000_000, 80_000_000, steps, t)
    assert (cp.num_steps, cp.length) == (4000, 1) and rx.num_steps == steps and cp.chirpness != rx.chirpness
    last = 0
    for c in range(nbuf):
        want = oracle_mod.chirp_gen(cp, last, L, 0.25)
        np.testing.assert_allclose(y[c * L:(c + 1) * L], want, rtol=0, atol=3e-7)
        last = (last + L) % (cp.num_steps * cp.length)


def test_tx_class_buffer_longer_than_a_second_and_noise_request(cuda_device, gsdr_lib, oracle_mod, tmp_path):
    """TONES with buffer_len > rate: the period buffer is rate * ceil(buffer_len / rate) samples plus one buffer
    (ref: cpp/USRP_buffer_generator.cpp:77-95) and get() walks it with TONES_last_sample wrapping at that length.
    A TX request of wave type NOISE generates the same comb: the reference's `case NOISE` falls through into
    TONES (:52-58)."""
    rate, L, nbuf = 50_000, 120_001, 4
    freq, ampl = [1000, -7000, 12_345], [0.3, 0.2, 0.1]
    outs = {}
    for mode in ("TONES", "NOISE"):
        info, y = run_tx(tmp_path, [f"mode {mode}", f"rate {rate}", f"buffer_len {L}", "channels 3",
                                    "freq " + " ".join(map(str, freq)), "ampl " + " ".join(map(str, ampl))], nbuf)
        assert y.size == nbuf * L
        outs[mode] = y
    period = rate * 3
    start = 0
    for c in range(nbuf):
        n = np.arange(start, start + L) % rate
        want = oracle_mod.tone_gen(freq, ampl, rate, 0, rate)[n]
        assert float(np.max(np.abs(outs["TONES"][c * L:(c + 1) * L] - want))) <= 2e-6 * sum(ampl), c
        start = (start + L) % period
    np.testing.assert_array_equal(outs["TONES"], outs["NOISE"])
