"""Regenerates the frozen regression vectors in this directory FROM THE ORACLE
(oracle/gsdr_oracle.c).  They are NOT reference outputs: the reference cannot
be built or imported in this image ("Oracle").  Purpose: freeze
the oracle's behaviour so that a later edit cannot silently move the target of
every GPU parity test.  Inputs are seeded; run `python tests/golden/make_golden.py`.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def crandn(rng, n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def run(dem, x, L):
    outs = [dem.process(x[c * L:(c + 1) * L]).ravel() for c in range(len(x) // L)]
    return np.concatenate(outs), np.array([len(o) for o in outs], dtype=np.int64)


def main():
    rng = np.random.default_rng(20251004)
    # DIRECT: 4 buffers of 400 at rate 1000 -> the NCO index wraps; negative tone included
    cfg = dict(rate=1000, decim=20, pf_average=4, buffer_len=400, freq=[0, 37, -211, 499, -500])
    x = crandn(rng, 4 * cfg["buffer_len"])
    y, n = run(oracle.Direct(cfg["freq"], cfg["rate"], cfg["decim"], cfg["pf_average"], cfg["buffer_len"]),
               x, cfg["buffer_len"])
    np.savez(os.path.join(HERE, "direct.npz"), config=json.dumps(cfg), x=x, y=y, lengths=n)
    # TONES: nfft does not divide L -> buffer_helper carry
    cfg = dict(rate=1200, fft_tones=12, pf_average=3, buffer_len=100, freq=[0, 100, -250, 433])
    x = crandn(rng, 5 * cfg["buffer_len"])
    y, n = run(oracle.Pfb(cfg["freq"], cfg["rate"], cfg["fft_tones"], cfg["pf_average"], cfg["buffer_len"]),
               x, cfg["buffer_len"])
    np.savez(os.path.join(HERE, "pfb.npz"), config=json.dumps(cfg), x=x, y=y, lengths=n)
    # CHIRP: ppt = 14 does not divide 500 -> VNA carry; sweep shorter than the data -> index wrap
    cfg = dict(rate=1000000, freq=-100000, chirp_f=100000, swipe_s=50, chirp_t=0.00035, decim=2,
               buffer_len=500)
    x = crandn(rng, 4 * cfg["buffer_len"])
    y, n = run(oracle.Chirp(cfg["rate"], cfg["freq"], cfg["chirp_f"], cfg["swipe_s"], cfg["chirp_t"],
                            cfg["decim"], cfg["buffer_len"]), x, cfg["buffer_len"])
    np.savez(os.path.join(HERE, "chirp.npz"), config=json.dumps(cfg), x=x, y=y, lengths=n)
    # NOISE (full spectrum): every bin of nfft = 12, carry as in TONES (round 2; drawn after the
    # three above so that those stay byte-identical)
    cfg = dict(fft_tones=12, pf_average=3, buffer_len=100)
    x = crandn(rng, 5 * cfg["buffer_len"])
    y, n = run(oracle.Noise(cfg["fft_tones"], cfg["pf_average"], cfg["buffer_len"]), x, cfg["buffer_len"])
    np.savez(os.path.join(HERE, "noise.npz"), config=json.dumps(cfg), x=x, y=y, lengths=n)
    # TX tone comb: 0 Hz and f = +rate dropped, f = -rate is DC, last duplicate wins; three
    # buffers of 400 samples wrap the length-1000 table
    cfg = dict(rate=1000, buffer_len=400, freq=[100, -250, 0, 77, 100, -1000, 1000, 500, -250, -1],
               ampl=[0.1, 0.2, 0.3, 0.05, 0.4, 0.07, 0.9, 0.11, 0.6, 0.02])
    y = np.concatenate([oracle.tone_gen(cfg["freq"], cfg["ampl"], cfg["rate"], (c * cfg["buffer_len"]) % cfg["rate"],
                                        cfg["buffer_len"]) for c in range(3)])
    np.savez(os.path.join(HERE, "tonegen.npz"), config=json.dumps(cfg), y=y)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
