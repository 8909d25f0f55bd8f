"""gpu_sdr_amd -- MI355X-native RX demodulation engine for the GPU_SDR readout.

Only what the hot path needs: csrc/ (HIP kernels + the C ABI of libgsdr.so),
the host-side mirror of the reference's demodulator interface, and the
synthetic IQ source used for benchmarking.
"""
from .demodulator import (GsdrError, RX_buffer_demodulator, RX_wrapper,  # noqa: F401
                          VNA_decimator_helper, buffer_helper, chirp_derive,
                          make_flat_window, make_sinc_window, param, pfb_batching,
                          pfb_tone_bins, string_to_w_type, w_type, w_type_to_str)

from .generator import TX_buffer_generator  # noqa: E402,F401

__all__ = [
    "TX_buffer_generator",
    "GsdrError", "RX_buffer_demodulator", "RX_wrapper", "VNA_decimator_helper",
    "buffer_helper", "chirp_derive", "make_flat_window", "make_sinc_window", "param",
    "pfb_batching", "pfb_tone_bins", "string_to_w_type", "w_type", "w_type_to_str",
]
