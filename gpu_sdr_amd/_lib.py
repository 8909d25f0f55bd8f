"""ctypes binding of libgsdr.so (include/gsdr.h).

The product path has NO CPU fallback: if the HIP library is missing this module
raises at import of the symbols, and every compute call needs a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSDR_LIB: load a differently built copy (timing-only ablation builds in scratch/)
LIB_PATH = os.environ.get("GSDR_LIB") or os.path.join(_HERE, "libgsdr.so")


class GsdrLibraryError(RuntimeError):
    pass


class ParamC(C.Structure):
    """struct gsdr_param_c (include/gsdr.h)"""
    _fields_ = [
        ("rate", C.c_int),
        ("decim", C.c_longlong),
        ("fft_tones", C.c_int),
        ("pf_average", C.c_longlong),
        ("buffer_len", C.c_longlong),
        ("wave_type", C.POINTER(C.c_int)),
        ("n_wave_type", C.c_int),
        ("freq", C.POINTER(C.c_int)),
        ("n_freq", C.c_int),
        ("chirp_t", C.POINTER(C.c_float)),
        ("n_chirp_t", C.c_int),
        ("chirp_f", C.POINTER(C.c_int)),
        ("n_chirp_f", C.c_int),
        ("swipe_s", C.POINTER(C.c_int)),
        ("n_swipe_s", C.c_int),
        ("device_index", C.c_int),
    ]


class BufferHelperC(C.Structure):
    """struct gsdr_buffer_helper"""
    _fields_ = [(n, C.c_int) for n in (
        "n_tones", "eff_length", "buffer_len", "average", "n_eff_tones",
        "new_0", "copy_size", "current_batch", "spare_samples", "spare_begin")]


class VnaHelperC(C.Structure):
    """struct gsdr_vna_helper"""
    _fields_ = [(n, C.c_int) for n in (
        "valid_size", "new0", "total_len", "spare_begin", "ppt", "buffer_len")]


class AntennaInfoC(C.Structure):
    """struct gsdr_antenna_info"""
    _fields_ = [("mode", C.c_int), ("rf", C.c_double), ("gain", C.c_int), ("bw", C.c_int),
                ("tuning_mode", C.c_int), ("samples", C.c_longlong), ("delay", C.c_double),
                ("burst_on", C.c_float), ("burst_off", C.c_float), ("data_mem_mult", C.c_longlong),
                ("ampl", C.POINTER(C.c_float)), ("n_ampl", C.c_int)]


class RxHeaderC(C.Structure):
    """struct gsdr_rx_header"""
    _fields_ = [("usrp_number", C.c_int), ("front_end_code", C.c_char), ("packet_number", C.c_int),
                ("length", C.c_int), ("errors", C.c_int), ("channels", C.c_int)]


class ChirpParamC(C.Structure):
    """struct gsdr_chirp_param"""
    _fields_ = [("num_steps", C.c_ulonglong), ("length", C.c_ulonglong),
                ("chirpness", C.c_uint), ("f0", C.c_int)]


# every symbol include/gsdr.h declares: (name, restype, argtypes)
_vp, _ip, _fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
SIGNATURES = [
    ("gsdr_demod_create", _vp, [C.POINTER(ParamC)]),
    ("gsdr_demod_process", C.c_int, [_vp, _vp, _vp]),
    ("gsdr_demod_process_device", C.c_int, [_vp, _vp, _vp, _vp]),
    ("gsdr_demod_submit", C.c_int, [_vp, _vp, _vp]),
    ("gsdr_demod_submit_device", C.c_int, [_vp, _vp, _vp]),
    ("gsdr_demod_wait", C.c_int, [_vp]),
    ("gsdr_demod_prepare", C.c_int, [_vp, C.c_int]),
    ("gsdr_demod_close", None, [_vp]),
    ("gsdr_last_error", C.c_char_p, [_vp]),
    ("gsdr_abi_version", C.c_int, []),
    ("gsdr_build_info", C.c_char_p, []),
    ("gsdr_reload_env", None, []),
    ("gsdr_demod_describe", C.c_int, [_vp, C.c_char_p, C.c_int]),
    ("gsdr_demod_mode", C.c_int, [_vp]),
    ("gsdr_demod_channels", C.c_int, [_vp]),
    ("gsdr_demod_out_capacity", C.c_longlong, [_vp]),
    ("gsdr_demod_fcut", C.c_float, [_vp]),
    ("gsdr_demod_get_window", C.c_int, [_vp, _fp, C.c_int]),
    ("gsdr_demod_get_bins", C.c_int, [_vp, _ip, C.c_int]),
    ("gsdr_demod_profile_enable", None, [_vp, C.c_int]),
    ("gsdr_demod_profile_read", C.c_int, [_vp, C.POINTER(C.c_double)]),
    ("gsdr_demod_kernel_name", C.c_char_p, [_vp]),
    ("gsdr_make_sinc_window", None, [C.c_int, C.c_float, _fp]),
    ("gsdr_make_flat_window", None, [C.c_int, C.c_int, _fp]),
    ("gsdr_buffer_helper_init", None, [C.POINTER(BufferHelperC)] + [C.c_int] * 4),
    ("gsdr_buffer_helper_update", None, [C.POINTER(BufferHelperC)]),
    ("gsdr_vna_helper_init", None, [C.POINTER(VnaHelperC), C.c_int, C.c_int]),
    ("gsdr_vna_helper_update", None, [C.POINTER(VnaHelperC)]),
    ("gsdr_pfb_tone_bins", None, [C.c_int, C.c_int, _ip, C.c_int, _ip]),
    ("gsdr_pfb_batching", C.c_int, [C.c_longlong, C.c_int, C.c_longlong]),
    ("gsdr_pfb_lds_stages", C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    ("gsdr_txgen_tones_create", C.c_void_p, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int]),
    ("gsdr_txgen_tones_fill", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p]),
    ("gsdr_txgen_close", None, [C.c_void_p]),
    ("gsdr_txgen_create", C.c_void_p, [C.POINTER(ParamC), C.POINTER(C.c_float), C.c_int]),
    ("gsdr_txgen_get", C.c_int, [C.c_void_p, C.c_void_p]),
    ("gsdr_txgen_get_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("gsdr_txgen_buffer_len", C.c_longlong, [C.c_void_p]),
    ("gsdr_txgen_get_ptr", C.c_void_p, [C.c_void_p]),
    ("gsdr_txgen_prepare_host", C.c_int, [C.c_void_p]),
    ("gsdr_txgen_mode", C.c_int, [C.c_void_p]),
    ("gsdr_chirp_derive", None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                 C.POINTER(ChirpParamC)]),
    ("gsdr_chirp_derive_tx", None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.POINTER(ChirpParamC)]),
    ("gsdr_command_parse", _vp, [C.c_char_p, C.c_int]),
    ("gsdr_command_free", None, [_vp]),
    ("gsdr_command_error", C.c_char_p, []),
    ("gsdr_command_device", C.c_int, [_vp]),
    ("gsdr_command_antenna", C.c_int, [_vp, C.c_int, C.POINTER(ParamC), C.POINTER(AntennaInfoC)]),
    ("gsdr_server_reply", C.c_int, [C.c_int, C.c_char_p, C.c_char_p, C.c_int]),
    ("gsdr_format_async_header", None, [C.c_int, C.POINTER(C.c_ubyte)]),
    ("gsdr_format_rx_header", None, [C.POINTER(RxHeaderC), C.POINTER(C.c_ubyte)]),
    ("gsdr_source_tones", C.c_int, [_vp, C.c_longlong, C.c_longlong, C.c_int, _ip, _fp, _fp,
                                    C.c_int, C.c_float, C.c_ulonglong, _vp]),
    ("gsdr_tx_tone_bins", C.c_int, [C.c_int, _ip, _fp, C.c_int, _ip, _fp]),
    ("gsdr_source_chirp", C.c_int, [_vp, C.c_longlong, C.c_ulonglong,
                                    C.POINTER(ChirpParamC), C.c_float, _vp]),
]

_lib = None


def lib() -> C.CDLL:
    """Load libgsdr.so (built in-tree by gpu_sdr_amd/csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as
    # the system one libgsdr.so is linked against).  Whichever copy is mapped
    # first serves both, and torch only works with its own: load torch first so
    # that device pointers, streams and this library share ONE runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise GsdrLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C gpu_sdr_amd/csrc`. There is no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise GsdrLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, restype, argtypes in SIGNATURES:
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            if os.environ.get("GSDR_LIB") and os.environ.get("GSDR_LIB_OLD_ABI"):
                continue      # scratch/ A/B runs against a library of an earlier round (bench.py marks such lines INVALID)
            raise GsdrLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L
