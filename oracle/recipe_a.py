"""Recipe A -- the reference's own offline CPU demodulation, restated for Python 3.

TEST / BASELINE INFRASTRUCTURE ONLY: imported by bench.py's cpu_baseline leg and by
tests/.  The product path (gpu_sdr_amd) never imports it.

The only offline demodulation the reference has is the loop of
``scripts/raw_data_analisys.py:55-68`` (ref = /root/reference): per tone

    demodulation_signal = conj(exp(1j*(2*pi*tone/rate*arange(len(Z) + pi*0.25*tone/rate))))
    l = min(len(demodulation_signal), len(Z)) - 1
    res = demodulation_signal[:l] * Z[:l]
    res = scipy.signal.decimate(res, decimation, ftype='fir')[100:-100]

(the script is Python 2 + h5py and cannot be imported here: SURVEY.md section 8c).  The
misplaced parenthesis of the original -- the phase offset ends up inside arange's length
-- and the ``- 1`` are kept: they only change the length by one sample.  pyUSRP spreads
its analysis loops over ``N_CORES = 10`` joblib workers (pyUSRP/USRP_low_level.py:44-45);
here the tones are spread over a fork()ed process pool whose size is reported.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import time

import numpy as np

_Z = None
_RATE = 0
_DECIM = 0


def demod_tone(Z: np.ndarray, tone: int, rate: int, decimation: int, trim: bool = True) -> np.ndarray:
    """One pass of the loop body, scripts/raw_data_analisys.py:56-66."""
    from scipy import signal
    demodulation_signal = np.conj(np.exp(1.j * (np.pi * 2. * tone / rate * np.arange(len(Z) + np.pi * 0.25 * tone / rate))))
    l = min(len(demodulation_signal), len(Z)) - 1
    res = demodulation_signal[:l] * Z[:l]
    if decimation:
        res = signal.decimate(res, decimation, ftype='fir')
        if trim:
            res = res[100:-100]
    return res


def _init(Z, rate, decim):
    global _Z, _RATE, _DECIM
    _Z, _RATE, _DECIM = Z, rate, decim


def _work(tone):
    r = demod_tone(_Z, tone, _RATE, _DECIM)
    return float(np.abs(r).mean()) if len(r) else 0.0


def host_cores() -> int:
    """Cores this process may use: scheduler affinity, capped by a cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def run(Z: np.ndarray, tones, rate: int, decimation: int, procs: int) -> dict:
    """All `tones` of one buffer `Z` over `procs` worker processes.  Must be called
    BEFORE the calling process initialises the GPU (the pool fork()s)."""
    tones = [int(t) for t in tones]
    procs = max(1, min(int(procs), len(tones)))
    t0 = time.perf_counter()
    if procs == 1:
        _init(Z, rate, decimation)
        chk = [_work(t) for t in tones]
    else:
        ctx = mp.get_context("fork")
        with ctx.Pool(procs, initializer=_init, initargs=(Z, rate, decimation)) as pool:
            chk = pool.map(_work, tones, chunksize=max(1, len(tones) // (procs * 4)))
    dt = time.perf_counter() - t0
    return dict(seconds=dt, tones=len(tones), samples=len(Z), procs=procs,
                msamples_per_s=len(Z) / dt / 1e6, tone_msamples_per_s=len(tones) * len(Z) / dt / 1e6,
                checksum=float(np.sum(chk)))
