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
    """class TX_buffer_generator, headers/USRP_buffer_generator.hpp -- the Python face of gsdr_txgen_*
    (include/gsdr.h; include/USRP_buffer_generator.hpp is the C++ one).

    ``get(out)`` fills a torch complex64 CUDA tensor (gsdr_txgen_get_device, asynchronous on the current or
    the given stream) or a numpy complex64 array (gsdr_txgen_get, as the reference's get() to host memory)
    of ``buffer_len`` samples with the next TX buffer; ``close()`` frees the device tables."""

    def __init__(self, init_parameters: param, device_index: int = 0):
        import ctypes as C
        from .demodulator import _carr
        p = self.parameters = init_parameters
        self.buffer_len = int(p.buffer_len)
        L = self._L = _lib.lib()
        keep = []
        pc = _lib.ParamC()
        pc.rate = int(p.rate)
        pc.decim = int(p.decim)
        pc.fft_tones = int(p.fft_tones)
        pc.pf_average = int(p.pf_average)
        pc.buffer_len = int(p.buffer_len)
        a, pc.wave_type = _carr([int(w) for w in p.wave_type], C.c_int); keep.append(a)
        pc.n_wave_type = len(p.wave_type)
        a, pc.freq = _carr([int(f) for f in p.freq], C.c_int); keep.append(a)
        pc.n_freq = len(p.freq)
        a, pc.chirp_t = _carr([float(f) for f in p.chirp_t], C.c_float); keep.append(a)
        pc.n_chirp_t = len(p.chirp_t)
        a, pc.chirp_f = _carr([int(f) for f in p.chirp_f], C.c_int); keep.append(a)
        pc.n_chirp_f = len(p.chirp_f)
        a, pc.swipe_s = _carr([int(f) for f in p.swipe_s], C.c_int); keep.append(a)
        pc.n_swipe_s = len(p.swipe_s)
        pc.device_index = int(device_index)
        ampl = np.ascontiguousarray(list(p.ampl) if p.ampl else [], dtype=np.float32)
        self._h = L.gsdr_txgen_create(C.byref(pc), ampl.ctypes.data_as(C.POINTER(C.c_float)), len(ampl))
        if not self._h:
            raise GsdrError(L.gsdr_last_error(None).decode())
        self.mode = w_type(L.gsdr_txgen_mode(self._h))      # a NOISE request is TONES (the reference falls through)

    def get_view(self) -> np.ndarray:
        """TONES as the reference's get() hands them out (get_from_tones, cpp/USRP_buffer_generator.cpp:226-229):
        the next buffer_len samples as a read-only VIEW of the generator's own host period buffer
        (gsdr_txgen_get_ptr); valid until close()."""
        import ctypes as C
        if not self._h:
            raise GsdrError("generator is closed")
        ptr = self._L.gsdr_txgen_get_ptr(self._h)
        if not ptr:
            raise GsdrError(self._L.gsdr_last_error(None).decode())
        buf = (C.c_float * (2 * self.buffer_len)).from_address(ptr)
        v = np.frombuffer(buf, dtype=np.complex64, count=self.buffer_len)
        v.flags.writeable = False
        return v

    def get(self, out, stream=None) -> None:
        import ctypes as C
        if not self._h:
            raise GsdrError("generator is closed")
        if isinstance(out, np.ndarray):
            if out.dtype != np.complex64 or not out.flags.c_contiguous or out.size < self.buffer_len:
                raise TypeError("need a contiguous complex64 array of buffer_len samples")
            rc = self._L.gsdr_txgen_get(self._h, out.ctypes.data)
        else:
            import torch
            if not (out.is_cuda and out.dtype == torch.complex64 and out.is_contiguous() and out.numel() >= self.buffer_len):
                raise TypeError("need a contiguous complex64 CUDA tensor of buffer_len samples")
            if stream is None:
                stream = torch.cuda.current_stream(out.device)
            rc = self._L.gsdr_txgen_get_device(self._h, out.data_ptr(), C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise GsdrError(self._L.gsdr_last_error(None).decode())

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.gsdr_txgen_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
