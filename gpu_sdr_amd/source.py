"""Synthetic in-memory IQ source: replaces the UHD hardware manager for
benchmarking and tests (shape of the reference's --sw_loop RX thread,
/root/reference/cpp/USRP_hardware_manager.cpp:1331-1395).

    x[n] = sum_k a_k exp(i(2 pi f_k (n mod rate)/rate + phi_k)) + sigma (g1 + i g2)

a_k = 1/N (pyUSRP/USRP_noise.py:477), f_k distinct integer Hz in
(-rate/2, rate/2) (scripts/get_noise.py:91), phi_k ~ U[0, 2 pi).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def tone_comb(n_tones: int, rate: int, seed: int):
    """Deterministic tone set: (freq[int32], ampl[f32], phase[f32])."""
    rng = np.random.default_rng(seed)
    half = rate // 2
    freq = set()
    while len(freq) < n_tones:
        cand = rng.integers(-half + 1, half, size=n_tones - len(freq))
        freq.update(int(c) for c in cand)
    freq = np.array(sorted(freq), dtype=np.int32)
    rng.shuffle(freq)
    ampl = np.full(n_tones, 1.0 / n_tones, dtype=np.float32)
    phase = rng.uniform(0.0, 2.0 * np.pi, size=n_tones).astype(np.float32)
    return freq, ampl, phase


def host_tones(n: int, start: int, rate: int, freq, ampl, phase, sigma: float = 0.0,
               seed: int = 0, tone_block: int = 64) -> np.ndarray:
    """Host (numpy, float64 phase) version of the source formula; for tests."""
    idx = (start + np.arange(n, dtype=np.int64)) % rate
    acc = np.zeros(n, dtype=np.complex128)
    freq = np.asarray(freq, dtype=np.int64)
    for k0 in range(0, len(freq), tone_block):
        f = freq[k0:k0 + tone_block, None]
        ph = (f * idx[None, :]) % rate
        ang = 2.0 * np.pi * ph / rate + np.asarray(phase[k0:k0 + tone_block], dtype=np.float64)[:, None]
        acc += (np.asarray(ampl[k0:k0 + tone_block], dtype=np.float64)[:, None] * np.exp(1j * ang)).sum(0)
    if sigma > 0:
        rng = np.random.default_rng(seed)
        acc += sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return acc.astype(np.complex64)


def device_tones(out_tensor, start: int, rate: int, freq, ampl, phase, sigma: float = 0.0,
                 seed: int = 0, stream=None) -> None:
    """Fill a torch complex64 CUDA tensor with the source signal (HIP kernel)."""
    import torch
    assert out_tensor.is_cuda and out_tensor.dtype == torch.complex64 and out_tensor.is_contiguous()
    freq = np.ascontiguousarray(freq, dtype=np.int32)
    ampl = np.ascontiguousarray(ampl, dtype=np.float32)
    phase = np.ascontiguousarray(phase, dtype=np.float32)
    if stream is None:
        stream = torch.cuda.current_stream(out_tensor.device)
    L = _lib.lib()
    with torch.cuda.device(out_tensor.device):
        rc = L.gsdr_source_tones(out_tensor.data_ptr(), out_tensor.numel(), int(start), int(rate),
                                 freq.ctypes.data_as(C.POINTER(C.c_int)),
                                 ampl.ctypes.data_as(C.POINTER(C.c_float)),
                                 phase.ctypes.data_as(C.POINTER(C.c_float)),
                                 len(freq), C.c_float(sigma), C.c_ulonglong(seed),
                                 C.c_void_p(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(L.gsdr_last_error(None).decode())


def device_chirp(out_tensor, last_index: int, cp, scale: float = 1.0, stream=None) -> None:
    """TX chirp law (ref: chirp_gen, cpp/kernels.cu:335-372) on the device."""
    import torch
    assert out_tensor.is_cuda and out_tensor.dtype == torch.complex64 and out_tensor.is_contiguous()
    if stream is None:
        stream = torch.cuda.current_stream(out_tensor.device)
    L = _lib.lib()
    with torch.cuda.device(out_tensor.device):
        rc = L.gsdr_source_chirp(out_tensor.data_ptr(), out_tensor.numel(), C.c_ulonglong(last_index),
                                 C.byref(cp), C.c_float(scale), C.c_void_p(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(L.gsdr_last_error(None).decode())
