"""CPU oracle for the GPU_SDR RX demodulation path -- TEST INFRASTRUCTURE ONLY.

"parity unpinned": see oracle/gsdr_oracle.h.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; gpu_sdr_amd must not.

Thin ctypes binding over oracle/liboracle.so (built by oracle/Makefile).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (no-op when it is newer than the source)."""
    src = os.path.join(_HERE, "gsdr_oracle.c")
    hdr = os.path.join(_HERE, "gsdr_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


class _BufferHelper(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_tones", "eff_length", "buffer_len", "average", "n_eff_tones",
        "new_0", "copy_size", "current_batch", "spare_samples", "spare_begin")]


class _VnaHelper(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "valid_size", "new0", "total_len", "spare_begin", "ppt", "buffer_len")]


class ChirpParam(C.Structure):
    _fields_ = [("num_steps", C.c_ulong), ("length", C.c_ulong),
                ("chirpness", C.c_uint), ("f0", C.c_int)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.oracle_make_sinc_window.argtypes = [C.c_int, C.c_float, fp]
    L.oracle_make_flat_window.argtypes = [C.c_int, C.c_int, fp]
    L.oracle_buffer_helper_init.argtypes = [C.POINTER(_BufferHelper)] + [C.c_int] * 4
    L.oracle_buffer_helper_update.argtypes = [C.POINTER(_BufferHelper)]
    L.oracle_vna_helper_init.argtypes = [C.POINTER(_VnaHelper), C.c_int, C.c_int]
    L.oracle_vna_helper_update.argtypes = [C.POINTER(_VnaHelper)]
    L.oracle_pfb_tone_bins.argtypes = [C.c_int, C.c_int, ip, C.c_int, ip]
    L.oracle_pfb_batching.argtypes = [C.c_long, C.c_int, C.c_long]
    L.oracle_pfb_batching.restype = C.c_int
    L.oracle_chirp_params.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.POINTER(ChirpParam)]
    L.oracle_direct_mix.argtypes = [ip, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp, vp]
    L.oracle_direct_create.argtypes = [ip, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long]
    L.oracle_direct_create.restype = vp
    L.oracle_direct_process.argtypes = [vp, vp, vp]
    L.oracle_direct_process.restype = C.c_long
    L.oracle_direct_destroy.argtypes = [vp]
    L.oracle_direct_taps.argtypes = [vp]
    L.oracle_direct_taps.restype = fp
    L.oracle_pfb_create.argtypes = [ip, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long]
    L.oracle_pfb_create.restype = vp
    L.oracle_noise_create.argtypes = [C.c_int, C.c_long, C.c_long]
    L.oracle_noise_create.restype = vp
    L.oracle_pfb_process.argtypes = [vp, vp, vp]
    L.oracle_pfb_process.restype = C.c_long
    L.oracle_pfb_destroy.argtypes = [vp]
    L.oracle_pfb_bins.argtypes = [vp]
    L.oracle_pfb_bins.restype = ip
    L.oracle_chirp_demod.argtypes = [C.POINTER(ChirpParam), C.c_ulong, C.c_size_t, vp, vp]
    L.oracle_chirp_gen.argtypes = [C.POINTER(ChirpParam), C.c_ulong, C.c_size_t, C.c_float, vp]
    L.oracle_chirp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_long, C.c_long]
    L.oracle_chirp_create.restype = vp
    L.oracle_chirp_process.argtypes = [vp, vp, vp]
    L.oracle_chirp_process.restype = C.c_long
    L.oracle_chirp_destroy.argtypes = [vp]
    L.oracle_tone_gen.argtypes = [ip, fp, C.c_int, C.c_int, C.c_float, C.c_long, C.c_size_t, vp]
    L.oracle_tone_gen.restype = C.c_int
    L.oracle_num_threads.restype = C.c_int
    L.oracle_set_num_threads.argtypes = [C.c_int]
    _lib = L
    return L


def _iarr(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.int32))
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a, a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))


def make_sinc_window(length: int, fc: float) -> np.ndarray:
    w = np.empty(length, dtype=np.float32)
    lib().oracle_make_sinc_window(length, C.c_float(fc), w.ctypes.data_as(C.POINTER(C.c_float)))
    return w


def make_flat_window(length: int, side: int) -> np.ndarray:
    w = np.empty(length, dtype=np.float32)
    lib().oracle_make_flat_window(length, side, w.ctypes.data_as(C.POINTER(C.c_float)))
    return w


class BufferHelper:
    """cpp/USRP_server_memory_management.cpp:104-156"""
    FIELDS = [n for n, _ in _BufferHelper._fields_]

    def __init__(self, n_tones, buffer_len, average, n_eff_tones):
        self._s = _BufferHelper()
        lib().oracle_buffer_helper_init(C.byref(self._s), n_tones, buffer_len, average, n_eff_tones)

    def update(self):
        lib().oracle_buffer_helper_update(C.byref(self._s))

    def __getattr__(self, k):
        if k in BufferHelper.FIELDS:
            return getattr(self._s, k)
        raise AttributeError(k)

    def state(self):
        return {k: getattr(self._s, k) for k in self.FIELDS}


class VnaHelper:
    """cpp/USRP_server_memory_management.cpp:30-56"""
    FIELDS = ["valid_size", "new0", "total_len", "spare_begin"]

    def __init__(self, ppt, buffer_len):
        self._s = _VnaHelper()
        lib().oracle_vna_helper_init(C.byref(self._s), ppt, buffer_len)

    def update(self):
        lib().oracle_vna_helper_update(C.byref(self._s))

    def __getattr__(self, k):
        if k in VnaHelper.FIELDS:
            return getattr(self._s, k)
        raise AttributeError(k)

    def state(self):
        return {k: getattr(self._s, k) for k in self.FIELDS}


def pfb_tone_bins(rate, fft_tones, freq) -> np.ndarray:
    f, fptr = _iarr(freq)
    bins = np.empty(len(f), dtype=np.int32)
    lib().oracle_pfb_tone_bins(rate, fft_tones, fptr, len(f), bins.ctypes.data_as(C.POINTER(C.c_int)))
    return bins


def pfb_batching(buffer_len, fft_tones, pf_average) -> int:
    return lib().oracle_pfb_batching(buffer_len, fft_tones, pf_average)


def chirp_params(rate, freq0, chirp_f, swipe_s, chirp_t) -> ChirpParam:
    cp = ChirpParam()
    lib().oracle_chirp_params(rate, freq0, chirp_f, swipe_s, C.c_float(chirp_t), C.byref(cp))
    return cp


def chirp_params_tx(rate, freq0, chirp_f, swipe_s, chirp_t) -> ChirpParam:
    """TX generator's derivation (cpp/USRP_buffer_generator.cpp:107-129)."""
    cp = ChirpParam()
    L = lib()
    L.oracle_chirp_params_tx.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(ChirpParam)]
    L.oracle_chirp_params_tx.restype = None
    L.oracle_chirp_params_tx(rate, freq0, chirp_f, swipe_s, C.c_float(chirp_t), C.byref(cp))
    return cp


def direct_mix(freq, rate, idx, x) -> np.ndarray:
    f, fptr = _iarr(freq)
    x, xp = _c64(x)
    out = np.empty((len(f), len(x)), dtype=np.complex64)
    lib().oracle_direct_mix(fptr, len(f), rate, idx, len(x), xp, out.ctypes.data_as(C.c_void_p))
    return out


class Direct:
    """DIRECT demodulator (cpp/USRP_demodulator.cpp:59-119, 400-464)."""

    def __init__(self, freq, rate, decim, pf_average, buffer_len):
        f, fptr = _iarr(freq)
        self.n_tones, self.decim, self.f, self.L = len(f), int(decim), int(pf_average), int(buffer_len)
        self._h = lib().oracle_direct_create(fptr, len(f), rate, decim, pf_average, buffer_len)
        if not self._h:
            raise ValueError("oracle_direct_create failed (buffer_len % decim != 0?)")

    def taps(self) -> np.ndarray:
        n = self.decim * self.f
        return np.ctypeslib.as_array(lib().oracle_direct_taps(self._h), shape=(n,)).copy()

    def process(self, x) -> np.ndarray:
        x, xp = _c64(x)
        assert len(x) == self.L
        rows = self.L // max(self.decim, 1)
        out = np.empty((rows, self.n_tones), dtype=np.complex64)
        n = lib().oracle_direct_process(self._h, xp, out.ctypes.data_as(C.c_void_p))
        assert n == rows * self.n_tones
        return out

    def close(self):
        if self._h:
            lib().oracle_direct_destroy(self._h)
            self._h = None

    __del__ = close


class Pfb:
    """TONES demodulator, decim == 0 (cpp/USRP_demodulator.cpp:121-175, 486-565)."""

    def __init__(self, freq, rate, fft_tones, pf_average, buffer_len):
        f, fptr = _iarr(freq)
        self.n_tones, self.nfft, self.avg, self.L = len(f), int(fft_tones), int(pf_average), int(buffer_len)
        self.batching = pfb_batching(buffer_len, fft_tones, pf_average)
        self._h = lib().oracle_pfb_create(fptr, len(f), rate, fft_tones, pf_average, buffer_len)

    def bins(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().oracle_pfb_bins(self._h), shape=(self.n_tones,)).copy()

    def process(self, x) -> np.ndarray:
        x, xp = _c64(x)
        assert len(x) == self.L
        out = np.zeros((self.batching, self.n_tones), dtype=np.complex64)
        n = lib().oracle_pfb_process(self._h, xp, out.ctypes.data_as(C.c_void_p))
        assert n % self.n_tones == 0
        return out[: n // self.n_tones]

    def close(self):
        if self._h:
            lib().oracle_pfb_destroy(self._h)
            self._h = None

    __del__ = close


class Noise(Pfb):
    """NOISE demodulator, decim == 0: full spectra (cpp/USRP_demodulator.cpp:264-313, 568-649)."""

    def __init__(self, fft_tones, pf_average, buffer_len):
        self.n_tones, self.nfft, self.avg, self.L = int(fft_tones), int(fft_tones), int(pf_average), int(buffer_len)
        self.batching = pfb_batching(buffer_len, fft_tones, pf_average)
        self._h = lib().oracle_noise_create(fft_tones, pf_average, buffer_len)


def chirp_demod(cp: ChirpParam, last_index: int, x) -> np.ndarray:
    x, xp = _c64(x)
    out = np.empty(len(x), dtype=np.complex64)
    lib().oracle_chirp_demod(C.byref(cp), last_index, len(x), xp, out.ctypes.data_as(C.c_void_p))
    return out


def chirp_gen(cp: ChirpParam, last_index: int, n: int, scale: float = 1.0) -> np.ndarray:
    out = np.empty(n, dtype=np.complex64)
    lib().oracle_chirp_gen(C.byref(cp), last_index, n, C.c_float(scale), out.ctypes.data_as(C.c_void_p))
    return out


def tone_gen(freq, ampl, rate: int, start: int, n: int, scale: float = 1.0) -> np.ndarray:
    """TX tone comb, samples [start, start+n) of the periodic length-`rate` buffer
    (cpp/kernels.cu:589-684, cpp/USRP_buffer_generator.cpp:226-229): 0 Hz / out-of-range
    tones are dropped, of equal frequencies the last wins -- see gsdr_oracle.h."""
    f, fptr = _iarr(freq)
    a = np.ascontiguousarray(np.asarray(ampl, dtype=np.float32))
    assert len(a) >= len(f)
    out = np.empty(n, dtype=np.complex64)
    used = lib().oracle_tone_gen(fptr, a.ctypes.data_as(C.POINTER(C.c_float)), len(f), int(rate),
                                 C.c_float(scale), int(start), n, out.ctypes.data_as(C.c_void_p))
    assert used >= 0
    return out


class Chirp:
    """CHIRP demodulator (cpp/USRP_demodulator.cpp:177-262, 342-397)."""

    def __init__(self, rate, freq0, chirp_f, swipe_s, chirp_t, decim, buffer_len):
        self.L, self.decim = int(buffer_len), int(decim)
        self._h = lib().oracle_chirp_create(rate, freq0, chirp_f, swipe_s, C.c_float(chirp_t),
                                            decim, buffer_len)
        if not self._h:
            raise ValueError("oracle_chirp_create failed")

    def process(self, x) -> np.ndarray:
        x, xp = _c64(x)
        assert len(x) == self.L
        out = np.empty(self.L, dtype=np.complex64)
        n = lib().oracle_chirp_process(self._h, xp, out.ctypes.data_as(C.c_void_p))
        return out[:n].copy()

    def close(self):
        if self._h:
            lib().oracle_chirp_destroy(self._h)
            self._h = None

    __del__ = close
