"""Host-side mirror of the reference's TX_buffer_generator, used as the
synthetic in-memory IQ source for loop-back runs (the reference's --sw_loop
memcpys TX buffers into the RX queue, cpp/USRP_hardware_manager.cpp:1071-1123,
1331-1395).

  * TONES: the reference builds a length-`rate` buffer by placing ampl[k] at bin
    freq[k] (rate + freq[k] for negative tones) of an UNNORMALISED inverse FFT
    (tone_gen, cpp/kernels.cu:589-684) and serves successive buffer_len slices
    of it, wrapping at `rate` (get_from_tones, cpp/USRP_buffer_generator.cpp:226-229).
    That buffer is the closed form  x[n] = sum_bins X[bin] exp(+2 pi i bin n / rate),
    n taken mod rate, which the HIP source kernel evaluates directly.  The bin vector is
    filled as the reference fills it (`tone_bins`): index = f if f > 0 else rate + f,
    ASSIGNED -- of tones on the same bin the last one wins, they do not add -- and an
    index outside [0, rate) is dropped: a 0 Hz tone lands on index `rate`, one element
    past the reference's allocation (undefined behaviour there, cpp/kernels.cu:622-628),
    so the transform never sees it and no DC term is generated.
  * CHIRP: chirp_gen law (cpp/kernels.cu:335-372) scaled by ampl[0], the running
    index wrapping at num_steps*length (get_from_chirp, :208-221).
Other wave types raise like the reference exits (:37-52).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .demodulator import GsdrError, chirp_derive, param, w_type
from .source import device_chirp, device_tones


def tone_bins(freq, ampl, rate: int):
    """(signed Hz, amplitudes) of the tones the reference's tone_gen really generates
    (gsdr_tx_tone_bins, ref: cpp/kernels.cu:617-635); see the module docstring for the quirks."""
    import ctypes as C
    f = np.ascontiguousarray(freq, dtype=np.int32)
    a = np.ascontiguousarray(ampl, dtype=np.float32)[: len(f)]
    of, oa = np.empty(len(f), dtype=np.int32), np.empty(len(f), dtype=np.float32)
    n = _lib.lib().gsdr_tx_tone_bins(int(rate), f.ctypes.data_as(C.POINTER(C.c_int)),
                                     a.ctypes.data_as(C.POINTER(C.c_float)), len(f),
                                     of.ctypes.data_as(C.POINTER(C.c_int)), oa.ctypes.data_as(C.POINTER(C.c_float)))
    if n < 0:
        raise GsdrError("gsdr_tx_tone_bins: bad arguments")
    return of[:n].copy(), oa[:n].copy()


class TX_buffer_generator:
    """class TX_buffer_generator, headers/USRP_buffer_generator.hpp.

    ``get(out)`` fills a torch complex64 CUDA tensor of ``buffer_len`` samples
    with the next TX buffer; ``close()`` is a no-op kept for symmetry."""

    def __init__(self, init_parameters: param):
        p = self.parameters = init_parameters
        self.buffer_len = int(p.buffer_len)
        if not p.wave_type:
            raise GsdrError("TX buffer generation needs at least one wave_type")
        last = p.wave_type[0]
        if sum(1 for w in p.wave_type if w == w_type.CHIRP) > 1:
            raise GsdrError("Multiple chirp TX buffer generation has been requested. "
                            "This feature is not implemented yet.")
        if any(w != last for w in p.wave_type):
            raise GsdrError("Mixed TX buffer generation has been requested. "
                            "This feature is not implemented yet.")
        self.mode = w_type(last)
        if self.mode in (w_type.NODSP, w_type.SWONLY):
            raise GsdrError("NODSP CASE NOT IMPLEMENTED.")
        if self.mode in (w_type.RAMP, w_type.DIRECT):
            raise GsdrError("RAMP CASE NOT IMPLEMENTED.")
        if self.mode == w_type.NOISE:
            raise GsdrError("NOISE TX generation is empty in the reference (get_from_noise)")
        if self.mode == w_type.TONES:
            n = len(p.wave_type)
            if len(p.freq) < n or len(p.ampl) < n:
                raise GsdrError("TONES needs freq[] and ampl[] for every wave_type entry")
            self._freq, self._ampl = tone_bins(p.freq[:n], p.ampl[:n], int(p.rate))
            self._phase = np.zeros(len(self._freq), dtype=np.float32)
            # TONES_buffer_len: rate, or the multiple of it that holds one buffer (:60-75)
            self._period = int(p.rate) * max(1, -(-self.buffer_len // int(p.rate)))
            self._last = 0            # TONES_last_sample
            # the device-side generator (gsdr_txgen_*): tables once, every buffer synthesised on demand
            import ctypes as C
            self._tx = None
            self._device = None
        else:  # CHIRP
            cp = chirp_derive(p.rate, p.freq[0], p.chirp_f[0], p.swipe_s[0], p.chirp_t[0])
            # the TX side also resets num_steps when a step would be shorter than one
            # sample (cpp/USRP_buffer_generator.cpp:111-115); the RX side does not
            if np.float32(p.chirp_t[0]) * np.float32(p.rate) / np.float32(cp.num_steps) < 1:
                cp.num_steps = int(np.float32(p.chirp_t[0]) * np.float32(p.rate))
            self._cp = cp
            self._scale = float(p.ampl[0]) if p.ampl else 1.0
            self._last = 0            # last_index

    def get(self, out_tensor, stream=None) -> None:
        if self.mode == w_type.TONES:
            import ctypes as C
            import torch
            assert out_tensor.is_cuda and out_tensor.dtype == torch.complex64 and out_tensor.is_contiguous()
            L = _lib.lib()
            if self._tx is None:
                f = np.ascontiguousarray(self._freq, dtype=np.int32)
                a = np.ascontiguousarray(self._ampl, dtype=np.float32)
                ph = np.ascontiguousarray(self._phase, dtype=np.float32)
                self._device = out_tensor.device
                self._tx = L.gsdr_txgen_tones_create(int(self.parameters.rate), f.ctypes.data_as(C.POINTER(C.c_int)),
                                                     a.ctypes.data_as(C.POINTER(C.c_float)), ph.ctypes.data_as(C.POINTER(C.c_float)),
                                                     len(f), self._device.index if self._device.index is not None else 0)
                if not self._tx:
                    raise GsdrError(L.gsdr_last_error(None).decode())
            if stream is None:
                stream = torch.cuda.current_stream(out_tensor.device)
            if L.gsdr_txgen_tones_fill(self._tx, out_tensor.data_ptr(), out_tensor.numel(), int(self._last),
                                       C.c_void_p(stream.cuda_stream)) != 0:
                raise GsdrError(L.gsdr_last_error(None).decode())
            self._last = (self._last + self.buffer_len) % self._period
        else:
            device_chirp(out_tensor, self._last, self._cp, scale=self._scale, stream=stream)
            self._last = (self._last + self.buffer_len) % (self._cp.num_steps * self._cp.length)

    def close(self) -> None:
        if getattr(self, "_tx", None):
            _lib.lib().gsdr_txgen_close(self._tx)
            self._tx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
