"""Host-side mirror of the reference's demodulator interface over the C ABI.

Same names and argument meaning as the C++ the server uses:
  * ``w_type`` / ``string_to_w_type``  -- headers/USRP_server_settings.hpp:114,
    cpp/USRP_server_settings.cpp:38-54
  * ``param``                          -- headers/USRP_server_settings.hpp:130-167
  * ``RX_wrapper``                     -- headers/USRP_server_settings.hpp:216-224
  * ``RX_buffer_demodulator``          -- headers/USRP_demodulator.hpp:13-33
(citations relative to /root/reference).

All arithmetic happens in libgsdr.so (HIP); this module only marshals
parameters and pointers.  torch is used for device memory and stream handles.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib


class GsdrError(RuntimeError):
    """Raised where the reference would print_error() and exit(-1), and on
    device errors (the reference ignores CUDA return codes)."""


class w_type(enum.IntEnum):
    """enum w_type, headers/USRP_server_settings.hpp:114"""
    TONES = 0
    CHIRP = 1
    NOISE = 2
    RAMP = 3
    NODSP = 4
    SWONLY = 5
    DIRECT = 6


def string_to_w_type(s: str) -> w_type:
    """cpp/USRP_server_settings.cpp:38-54: unknown strings (and "RAMP") map to NODSP."""
    return {"NODSP": w_type.NODSP, "CHIRP": w_type.CHIRP, "NOISE": w_type.NOISE,
            "TONES": w_type.TONES, "SWONLY": w_type.SWONLY, "DIRECT": w_type.DIRECT}.get(
                s, w_type.NODSP)


def w_type_to_str(w: w_type) -> str:
    """cpp/USRP_server_settings.cpp:9-36"""
    try:
        return w_type(w).name
    except ValueError:
        return "UNINIT"


@dataclass
class param:
    """struct param, headers/USRP_server_settings.hpp:130-167 (RX-relevant defaults
    follow the client's, pyUSRP/USRP_files.py:449-478)."""
    mode: str = "OFF"
    rate: int = 0
    gain: int = 0
    bw: int = 0
    tone: int = 0
    samples: int = 0
    delay: float = 1.0
    burst_on: float = 0.0
    burst_off: float = 0.0
    buffer_len: int = 1000000
    tuning_mode: bool = True
    freq: List[int] = field(default_factory=list)
    wave_type: List[w_type] = field(default_factory=list)
    ampl: List[float] = field(default_factory=list)
    decim: int = 0
    chirp_t: List[float] = field(default_factory=list)
    chirp_f: List[int] = field(default_factory=list)
    swipe_s: List[int] = field(default_factory=list)
    data_mem_mult: int = 1
    fft_tones: int = 0
    pf_average: int = 4


@dataclass
class RX_wrapper:
    """struct RX_wrapper, headers/USRP_server_settings.hpp:216-224"""
    buffer: object = None
    usrp_number: int = 0
    front_end_code: str = "A"
    packet_number: int = 0
    length: int = 0
    errors: int = 0
    channels: int = 0


def _carr(values, ctype):
    arr = (ctype * max(len(values), 1))(*values)
    return arr, C.cast(arr, C.POINTER(ctype))


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class RX_buffer_demodulator:
    """class RX_buffer_demodulator, headers/USRP_demodulator.hpp:13-33.

    ``process(in, out)`` returns the number of valid complex samples written to
    ``out`` (all channels interleaved [sample][channel]); ``close()`` releases
    the device state.  ``in``/``out`` are numpy complex64 arrays (host path,
    synchronous like the reference) or torch complex64 CUDA tensors (device
    path, enqueued on the current torch stream, not synchronised).
    """

    def __init__(self, init_parameters: param, init_diagnostic: bool = False,
                 device_index: int = -1):
        self.parameters = init_parameters
        self.diagnostic = bool(init_diagnostic)
        L = _lib.lib()
        p = init_parameters
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
        self._L = L
        self._h = L.gsdr_demod_create(C.byref(pc))
        if not self._h:
            raise GsdrError(L.gsdr_last_error(None).decode())
        self.fcut = float(L.gsdr_demod_fcut(self._h))
        if self.diagnostic and self.mode in (w_type.TONES, w_type.NOISE):
            # ref: make_sinc_window(..., diagnostic, ...) dumps the window,
            # cpp/kernels.cu:290-296 (float2 records, imag = 0)
            w = self.window()
            rec = np.zeros((len(w), 2), dtype=np.float32)
            rec[:, 0] = w
            rec.tofile("USRP_polyphase_filter_window.dat")

    # -- introspection ------------------------------------------------------
    @property
    def mode(self) -> w_type:
        return w_type(self._L.gsdr_demod_mode(self._h))

    @property
    def channels(self) -> int:
        return self._L.gsdr_demod_channels(self._h)

    @property
    def out_capacity(self) -> int:
        return int(self._L.gsdr_demod_out_capacity(self._h))

    @property
    def kernel_name(self) -> str:
        return self._L.gsdr_demod_kernel_name(self._h).decode()

    def prepare(self, host: bool = True, pipeline: bool = True, pipeline_host: bool = True, rehearse: bool = True) -> None:
        """gsdr_demod_prepare: create now what the entries would create on first use; `rehearse` also runs a
        throw-away twin through a few buffers of zeros (the process-wide first-use costs of kernels, pinned
        copies and streams: 5 - 7 ms each, otherwise paid by the first packets)."""
        what = (1 if host else 0) | (2 if pipeline else 0) | (4 if pipeline_host else 0) | (8 if rehearse else 0)
        if self._L.gsdr_demod_prepare(self._h, what) != 0:
            raise GsdrError(self._L.gsdr_last_error(self._h).decode())

    def describe(self) -> dict:
        """The engine this handle resolved to (gsdr_demod_describe): dominant kernel, family,
        row tiles per workgroup, pipeline streams, every GSDR_* variable set in the process."""
        import json
        buf = C.create_string_buffer(4096)
        self._L.gsdr_demod_describe(self._h, buf, len(buf))
        return json.loads(buf.value.decode())

    def window(self) -> np.ndarray:
        n = self._L.gsdr_demod_get_window(self._h, None, 0)
        w = np.empty(n, dtype=np.float32)
        self._L.gsdr_demod_get_window(self._h, w.ctypes.data_as(C.POINTER(C.c_float)), n)
        return w

    def bins(self) -> np.ndarray:
        n = self._L.gsdr_demod_get_bins(self._h, None, 0)
        b = np.empty(n, dtype=np.int32)
        self._L.gsdr_demod_get_bins(self._h, b.ctypes.data_as(C.POINTER(C.c_int)), n)
        return b

    # -- the hot path ---------------------------------------------------------
    def process(self, in_buffer, out_buffer) -> int:
        if not self._h:
            raise GsdrError("demodulator is closed")
        if _is_torch(in_buffer) or _is_torch(out_buffer):
            return self.process_device(in_buffer, out_buffer)
        if in_buffer.dtype != np.complex64 or out_buffer.dtype != np.complex64:
            raise TypeError("buffers must be complex64 (float2)")
        if in_buffer.size < self.parameters.buffer_len:
            raise ValueError("input buffer shorter than parameters.buffer_len")
        if out_buffer.size < self.out_capacity:
            raise ValueError(f"output buffer needs room for {self.out_capacity} samples")
        if not (in_buffer.flags.c_contiguous and out_buffer.flags.c_contiguous):
            raise ValueError("buffers must be contiguous")
        n = self._L.gsdr_demod_process(self._h, in_buffer.ctypes.data, out_buffer.ctypes.data)
        if n < 0:
            raise GsdrError(self._L.gsdr_last_error(self._h).decode())
        return n

    def process_device(self, in_tensor, out_tensor, stream=None) -> int:
        """Device-pointer entry (gsdr_demod_process_device); asynchronous."""
        import torch
        if not self._h:
            raise GsdrError("demodulator is closed")
        for t in (in_tensor, out_tensor):
            if not (t.is_cuda and t.dtype == torch.complex64 and t.is_contiguous()):
                raise TypeError("need contiguous complex64 CUDA tensors")
        if in_tensor.numel() < self.parameters.buffer_len:
            raise ValueError("input tensor shorter than parameters.buffer_len")
        if out_tensor.numel() < self.out_capacity:
            raise ValueError(f"output tensor needs room for {self.out_capacity} samples")
        if stream is None:
            stream = torch.cuda.current_stream(in_tensor.device)
        n = self._L.gsdr_demod_process_device(self._h, in_tensor.data_ptr(), out_tensor.data_ptr(),
                                              C.c_void_p(stream.cuda_stream))
        if n < 0:
            raise GsdrError(self._L.gsdr_last_error(self._h).decode())
        return n

    def submit(self, in_buffer: np.ndarray, out_buffer: np.ndarray) -> None:
        """Pipelined host-pointer entry (gsdr_demod_submit): returns at once; the
        buffers (ideally pinned) must stay alive until the matching wait()."""
        if in_buffer.dtype != np.complex64 or out_buffer.dtype != np.complex64:
            raise TypeError("buffers must be complex64 (float2)")
        if in_buffer.size < self.parameters.buffer_len or out_buffer.size < self.out_capacity:
            raise ValueError("buffer too small")
        if self._L.gsdr_demod_submit(self._h, in_buffer.ctypes.data, out_buffer.ctypes.data) != 0:
            raise GsdrError(self._L.gsdr_last_error(self._h).decode())

    def submit_device(self, in_tensor, out_tensor) -> None:
        """Pipelined device-pointer entry (gsdr_demod_submit_device): in_tensor must be
        complete (synchronise its producer first), out_tensor distinct per outstanding call."""
        import torch
        if not self._h:
            raise GsdrError("demodulator is closed")
        for t in (in_tensor, out_tensor):
            if not (t.is_cuda and t.dtype == torch.complex64 and t.is_contiguous()):
                raise TypeError("need contiguous complex64 CUDA tensors")
        if in_tensor.numel() < self.parameters.buffer_len or out_tensor.numel() < self.out_capacity:
            raise ValueError("tensor too small")
        if self._L.gsdr_demod_submit_device(self._h, in_tensor.data_ptr(), out_tensor.data_ptr()) != 0:
            raise GsdrError(self._L.gsdr_last_error(self._h).decode())

    def wait(self) -> int:
        """Valid length of the oldest submitted buffer (gsdr_demod_wait)."""
        n = self._L.gsdr_demod_wait(self._h)
        if n == -2:
            raise GsdrError("nothing outstanding")
        if n < 0:
            raise GsdrError(self._L.gsdr_last_error(self._h).decode())
        return n

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.gsdr_demod_close(self._h)
            self._h = None

    # -- kernel timing (hipEvents on the launch stream) ----------------------
    def profile_enable(self, on=True) -> None:
        """on = True / 1: time every launch; an integer n > 1: every n-th launch."""
        self._L.gsdr_demod_profile_enable(self._h, int(on))

    def profile_read(self):
        ms = C.c_double(0.0)
        n = self._L.gsdr_demod_profile_read(self._h, C.byref(ms))
        return n, ms.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host-side helpers of the path, straight from the library --------------

def make_sinc_window(length: int, fc: float) -> np.ndarray:
    w = np.empty(length, dtype=np.float32)
    _lib.lib().gsdr_make_sinc_window(length, C.c_float(fc), w.ctypes.data_as(C.POINTER(C.c_float)))
    return w


def make_flat_window(length: int, side: int) -> np.ndarray:
    w = np.empty(length, dtype=np.float32)
    _lib.lib().gsdr_make_flat_window(length, side, w.ctypes.data_as(C.POINTER(C.c_float)))
    return w


class buffer_helper:
    """class buffer_helper, headers/USRP_server_memory_management.hpp:77-101"""
    FIELDS = [n for n, _ in _lib.BufferHelperC._fields_]

    def __init__(self, n_tones, buffer_len, average, n_eff_tones):
        self._s = _lib.BufferHelperC()
        _lib.lib().gsdr_buffer_helper_init(C.byref(self._s), n_tones, buffer_len, average, n_eff_tones)

    def update(self):
        _lib.lib().gsdr_buffer_helper_update(C.byref(self._s))

    def __getattr__(self, k):
        if k in buffer_helper.FIELDS:
            return getattr(self._s, k)
        raise AttributeError(k)

    def state(self):
        return {k: getattr(self._s, k) for k in self.FIELDS}


class VNA_decimator_helper:
    """class VNA_decimator_helper, headers/USRP_server_memory_management.hpp:24-40"""
    FIELDS = ["valid_size", "new0", "total_len", "spare_begin"]

    def __init__(self, init_ppt, init_buffer_len):
        self._s = _lib.VnaHelperC()
        _lib.lib().gsdr_vna_helper_init(C.byref(self._s), init_ppt, init_buffer_len)

    def update(self):
        _lib.lib().gsdr_vna_helper_update(C.byref(self._s))

    def __getattr__(self, k):
        if k in VNA_decimator_helper.FIELDS:
            return getattr(self._s, k)
        raise AttributeError(k)

    def state(self):
        return {k: getattr(self._s, k) for k in self.FIELDS}


def pfb_tone_bins(rate, fft_tones, freq) -> np.ndarray:
    f = np.ascontiguousarray(np.asarray(freq, dtype=np.int32))
    bins = np.empty(len(f), dtype=np.int32)
    ip = C.POINTER(C.c_int)
    _lib.lib().gsdr_pfb_tone_bins(rate, fft_tones, f.ctypes.data_as(ip), len(f), bins.ctypes.data_as(ip))
    return bins


def pfb_batching(buffer_len, fft_tones, pf_average) -> int:
    return _lib.lib().gsdr_pfb_batching(buffer_len, fft_tones, pf_average)


def chirp_derive(rate, freq0, chirp_f, swipe_s, chirp_t) -> _lib.ChirpParamC:
    cp = _lib.ChirpParamC()
    _lib.lib().gsdr_chirp_derive(rate, freq0, chirp_f, swipe_s, C.c_float(chirp_t), C.byref(cp))
    return cp


def chirp_derive_tx(rate, freq0, chirp_f, swipe_s, chirp_t) -> _lib.ChirpParamC:
    """The TX generator's own derivation (ref: cpp/USRP_buffer_generator.cpp:107-129)."""
    cp = _lib.ChirpParamC()
    _lib.lib().gsdr_chirp_derive_tx(rate, freq0, chirp_f, swipe_s, C.c_float(chirp_t), C.byref(cp))
    return cp
