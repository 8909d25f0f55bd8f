"""Recipe B -- a SECOND, independent CPU restatement of the RX demodulation path.

TEST INFRASTRUCTURE ONLY (same rule as oracle/gsdr_oracle.c: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Why it exists: the reference ships no fixtures and can be neither built nor
imported in this image, so nothing reference-held pins the C oracle ("parity
unpinned").  This file is the second opinion BASELINE.md
section 3 / SURVEY.md section 8(d) C1 planned: written from the reference
source lines cited below -- NOT from gsdr_oracle.c -- in vectorised numpy, and
it mirrors the reference's *mechanics* (its device buffers, the cuBLAS calls in
their column-major reading, the in-place carry moves) where the C oracle uses
closed forms over a global sample counter.  tests/test_recipe_b.py compares the
two on BASELINE config 1 in full and on PFB / chirp shapes; agreement of two
independently written restatements is the evidence that can exist here.

Arithmetic types follow the reference: float32 data, double sincos, complex64
cuBLAS products (numpy's cgemm accumulates in float32 like cuBLAS; pass
``acc=np.complex128`` for an fp64 accumulate when used as an arbiter).

All "ref" citations are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np

PI_F = np.float32(3.14159265358979)          # ref: headers/kernels.cuh:34


# --------------------------------------------------------------------------
# windows
# --------------------------------------------------------------------------
def make_sinc_window(length: int, fc: float) -> np.ndarray:
    """ref: cpp/kernels.cu:258-310 (real part; the imaginary part is 0).

    Expression types as the C++ evaluates them: ``2.f*pi_f*fc*sinc_index`` is float
    left to right, ``sin``/``cos`` of a float argument resolve to the float overloads,
    ``0.54-0.46*cos(..)`` is double, the product is narrowed to float, ``scale`` is a
    float accumulator in loop order, the division is float."""
    f32 = np.float32
    fc = f32(fc)
    i = np.arange(length, dtype=np.int64)
    sinc_index = i - (length - 1) // 2                    # :268, integer division
    arg = ((f32(2.0) * PI_F) * fc) * sinc_index.astype(f32)
    safe = np.where(sinc_index != 0, arg, f32(1.0))
    sinc = ((f32(2.0) * fc) * np.sin(safe, dtype=f32)) / safe          # :274
    sinc = np.where(sinc_index != 0, sinc, f32(2.0) * fc).astype(f32)  # :275
    ham_arg = ((f32(2.0) * PI_F) * i.astype(f32)) / f32(length - 1)    # :278
    ham = 0.54 - 0.46 * np.cos(ham_arg, dtype=f32).astype(np.float64)
    w = (sinc.astype(np.float64) * ham).astype(f32)
    scale = np.cumsum(w, dtype=f32)[-1]                   # :280, sequential float sum
    return (w / scale).astype(f32)                        # :285


def make_flat_window(length: int, side: int) -> np.ndarray:
    """ref: cpp/kernels.cu:208-253: zeros on [0, side), then the fill loop writes
    [side, length) (overwriting the trailing zeros), normalised by its float sum."""
    w = np.zeros(length, dtype=np.float32)
    w[side:length] = 1.0
    scale = np.float32(length - side)                     # sum of (length-side) ones, exact
    return (w / scale).astype(np.float32)


# --------------------------------------------------------------------------
# DIRECT: NCO mix + per-tone FIR + transpose
# --------------------------------------------------------------------------
def nco_mix(x: np.ndarray, freq, phases, rate: int, index_counter: int) -> np.ndarray:
    """ref: direct_demodulator_integer, cpp/kernels.cu:45-86.  Returns [N][L] complex64
    (tone-major, as the kernel writes ``output[ch*L + j]``)."""
    x = np.asarray(x, dtype=np.complex64)
    L = x.shape[0]
    tf = np.asarray(freq, dtype=np.int64)[:, None]
    tp = np.asarray(phases, dtype=np.int64)[:, None]
    ii = ((np.arange(L, dtype=np.int64) + np.int64(index_counter)) % np.int64(rate))[None, :]  # :65
    # C++ % truncates toward zero: the remainder has the sign of tf*ii (np.fmod does that)
    my_phase = tp + np.fmod(tf * ii, np.int64(rate))                   # :67
    ph = 2.0 * (my_phase.astype(np.float64) / float(rate))             # :68
    _q = np.sin(np.pi * ph)                                            # sincospi(ph, &_q, &_i) :71
    _i = np.cos(np.pi * ph)
    xr = x.real.astype(np.float64)[None, :]
    xi = x.imag.astype(np.float64)[None, :]
    out = np.empty((tf.shape[0], L), dtype=np.complex64)
    out.imag = (xi * _i - xr * _q).astype(np.float32)                  # :81
    out.real = (xr * _i + xi * _q).astype(np.float32)                  # :82
    return out


class Fir:
    """ref: class FIR, cpp/fir.cu:15-88, headers/fir.hpp:7-37 -- one instance per tone.
    `_dout` starts as zeros (the constructor's cudaMemset at :26 takes the address of
    the pointer; the intended semantics are zeros)."""

    def __init__(self, hcoeff: np.ndarray, M: int, f: int, nt: int, acc=np.complex64):
        assert nt % M == 0                                             # :20
        self.M, self.f, self.nb = M, f, nt // M
        self.acc = acc
        # _dcoeff is f x M complex with zero imaginary part; as the GEMM's B operand
        # (column-major, ldb = M) it is the M x f matrix B[i, c] = coeff[c*M + i]
        self.B = np.ascontiguousarray(np.asarray(hcoeff, dtype=np.float32).reshape(f, M).T).astype(acc)
        self.dout = np.zeros(self.nb + f - 1, dtype=acc)

    def run_fir(self, din: np.ndarray) -> np.ndarray:
        """fir_apply + fir_to_dev + fir_shift (:44-88)."""
        nb, f, M = self.nb, self.f, self.M
        # cublasCgemm(OP_T, OP_N, nb, f, M, 1, din, M, coeff, M, 0, trapz, nb) (:48-54):
        # A (column-major, lda = M) is M x nb with A[i, j] = din[j*M + i]; op(A) = A^T
        opA = np.asarray(din).reshape(nb, M).astype(self.acc, copy=False)
        trapz = opA @ self.B                                           # [nb][f]: C[j, c]
        for i in range(f):                                             # cublasCaxpy :56-61
            self.dout[f - i - 1: f - i - 1 + nb] += trapz[:, i]
        hout = self.dout[:nb].astype(np.complex64)                     # fir_to_dev :83
        rem = f - 1                                                    # fir_shift :64-69
        self.dout[:rem] = self.dout[nb: nb + rem].copy()
        self.dout[rem: rem + nb] = 0
        return hout


class Direct:
    """ref: RX_buffer_demodulator DIRECT case, cpp/USRP_demodulator.cpp:59-119 (ctor),
    :400-464 (process_direct)."""

    def __init__(self, freq, rate: int, decim: int, pf_average: int, buffer_len: int,
                 acc=np.complex64, tone_chunk: int = 8):
        self.freq = [int(v) for v in freq]                             # :66 int32 Hz
        self.phases = [0] * len(self.freq)                             # :67
        self.rate, self.decim, self.L = int(rate), int(decim), int(buffer_len)
        self.index = 0                                                 # DIRECT_current_index :88
        self.tone_chunk = tone_chunk
        if self.decim > 0:
            ntaps = self.decim * int(pf_average)
            # :99 make_sinc_window(decim*pf_average, 0.75/(decim*2.), false, true): the
            # double cut-off narrows to the float parameter
            self.taps = make_sinc_window(ntaps, np.float32(0.75 / (self.decim * 2.0)))
            self.fir = [Fir(self.taps, self.decim, int(pf_average), self.L, acc) for _ in self.freq]  # :110

    def process(self, x: np.ndarray) -> np.ndarray:
        """One buffer; returns [samples][channels] complex64 (the Cgeam transpose :422-433,
        :444-455)."""
        N = len(self.freq)
        n_out = self.L // max(self.decim, 1)                           # :402
        fir_output = np.empty((N, n_out), dtype=np.complex64)
        for c0 in range(0, N, self.tone_chunk):
            c1 = min(N, c0 + self.tone_chunk)
            mixed = nco_mix(x, self.freq[c0:c1], self.phases[c0:c1], self.rate, self.index)  # :407-417
            if self.decim > 0:
                for k in range(c0, c1):
                    fir_output[k] = self.fir[k].run_fir(mixed[k - c0])                        # :421
            else:
                fir_output[c0:c1] = mixed
        self.index = (self.index + self.L) % self.rate                 # :437-440
        return np.ascontiguousarray(fir_output.T)


# --------------------------------------------------------------------------
# TONES: polyphase filter + FFT + tone select, with the reference's buffer moves
# --------------------------------------------------------------------------
class BufferHelper:
    """ref: class buffer_helper, cpp/USRP_server_memory_management.cpp:104-156."""

    def __init__(self, n_tones: int, buffer_len: int, average: int, n_eff_tones: int):
        self.n_tones, self.buffer_len, self.average, self.n_eff_tones = n_tones, buffer_len, average, n_eff_tones
        self.eff_length = buffer_len
        self.current_batch = self.simulate_batching()
        self.spare_samples = self.eff_length - self.current_batch * n_tones
        self.spare_begin = self.eff_length - self.spare_samples
        self.new_0 = 0
        self.copy_size = n_eff_tones * self.current_batch

    def simulate_batching(self) -> int:                                # :145-156
        offset = batching = 0
        while offset + self.average * self.n_tones < self.eff_length:
            offset += self.n_tones
            batching += 1
        return batching

    def update(self) -> None:                                          # :125-142
        self.new_0 = self.spare_samples
        self.eff_length = self.spare_samples + self.buffer_len
        self.current_batch = self.simulate_batching()
        self.copy_size = self.n_eff_tones * self.current_batch
        self.spare_samples = self.eff_length - self.current_batch * self.n_tones
        self.spare_begin = self.eff_length - self.spare_samples


def pfb_tone_bins(rate: int, fft_tones: int, freq) -> list:
    """ref: upload_multitone_parameters, cpp/USRP_demodulator.cpp:722-733: every axis
    point within one bin width of the tone assigns; the LAST match wins.  -1 = no match
    (the reference leaves the entry uninitialised)."""
    bins = [-1] * len(freq)
    bin_size = float(rate) / float(fft_tones)
    for i in range(fft_tones):
        c = i * bin_size - bin_size * (fft_tones // 2)
        for u, f in enumerate(freq):
            if (f < c + bin_size) and (f > c - bin_size):
                bins[u] = (i + fft_tones // 2) % fft_tones
    return bins


class Pfb:
    """ref: TONES case, cpp/USRP_demodulator.cpp:121-175 (ctor), :486-565 (process_pfb,
    decim == 0 branch), kernels: polyphase_filter cpp/kernels.cu:474-516, cufftExecC2C
    (forward, batched, contiguous frames of nfft: numpy.fft.fft), tone_select :531-554."""

    def __init__(self, freq, rate: int, fft_tones: int, pf_average: int, buffer_len: int,
                 bins=None):
        self.nfft, self.avg, self.L = int(fft_tones), int(pf_average), int(buffer_len)
        self.fcut = np.float32(1.0 / (2 * self.nfft))                  # :131 (float member)
        self.window = make_sinc_window(self.nfft * self.avg, self.fcut)  # :134
        # :706 std::ceil((float)buffer_len/(float)fft_tones) + pf_average + 5
        self.batching = int(np.ceil(np.float32(self.L) / np.float32(self.nfft))) + self.avg + 5
        self.bins = list(bins) if bins is not None else pfb_tone_bins(rate, self.nfft, [int(f) for f in freq])
        self.n_eff = len(self.bins)
        self.raw_input = np.zeros(self.nfft * self.batching, dtype=np.complex64)   # :143
        self.buf = BufferHelper(self.nfft, self.L, self.avg, self.n_eff)          # :159

    def _polyphase_filter(self) -> np.ndarray:
        """kernels.cu:474-516 over the whole device buffer: offsets that do not have
        `average` frames behind them stay unwritten (zeros here; never selected)."""
        n, A, B = self.nfft, self.avg, self.batching
        total = B * n
        out = np.zeros(total, dtype=np.complex64)
        valid = max(0, total - n * A)                                  # offset + n*A < total  (:487)
        acc = np.zeros(valid, dtype=np.complex64)
        off = np.arange(valid)
        for i in range(A):                                             # float accumulate in loop order
            wi = self.window[(off % n) + i * n]
            acc = acc + self.raw_input[off + i * n] * wi
        out[:valid] = acc
        return out

    def process(self, x: np.ndarray) -> np.ndarray:
        b = self.buf
        self.raw_input[b.new_0: b.new_0 + self.L] = np.asarray(x, dtype=np.complex64)   # :491-495
        filtered = self._polyphase_filter()                                              # :498
        spectra = np.fft.fft(filtered.reshape(self.batching, self.nfft), axis=1)         # :501
        # move_buffer(raw_input, raw_input, spare_samples, spare_begin, 0) :504-509
        self.raw_input[: b.spare_samples] = self.raw_input[b.spare_begin: b.spare_begin + b.spare_samples].copy()
        cb = b.current_batch
        sel = spectra[:cb][:, np.asarray(self.bins, dtype=np.int64)]   # tone_select :537-550
        b.update()                                                      # :552
        return np.ascontiguousarray(sel).astype(np.complex64)          # [current_batch][n_eff]


# --------------------------------------------------------------------------
# CHIRP: demodulator + lock-in decimator
# --------------------------------------------------------------------------
def chirp_params(rate: int, freq0: int, chirp_f: int, swipe_s: int, chirp_t: float):
    """ref: CHIRP case, cpp/USRP_demodulator.cpp:192-214.  Returns (num_steps, length,
    chirpness, f0) with the struct's types (unsigned long, unsigned long, unsigned int, int)."""
    f32 = np.float32
    num_steps = int(swipe_s)
    if num_steps < 1:
        num_steps = int(f32(chirp_t) * f32(rate))                      # :194 float * int -> float
    length = int((f32(chirp_t) * f32(rate)) / f32(num_steps))          # :201 float math, truncated
    if length < 1:
        length = 1
    two32m1 = float(2 ** 32 - 1)
    chirpness_d = (two32m1 * (int(chirp_f) - int(freq0)) / (float(num_steps) - 1.0)) / float(rate)   # :210
    chirpness = int(chirpness_d) & 0xFFFFFFFF                          # double -> unsigned int
    f0 = int(two32m1 * (float(freq0) / float(rate)))                   # :214 double -> int (toward zero)
    return num_steps, length, chirpness, f0


def chirp_params_tx(rate: int, freq0: int, chirp_f: int, swipe_s: int, chirp_t: float):
    """ref: TX_buffer_generator CHIRP case, cpp/USRP_buffer_generator.cpp:107-129: as chirp_params, but a step
    shorter than one sample resets num_steps too (:118-122), and the slope (:125) uses the reset value."""
    f32 = np.float32
    num_steps = int(swipe_s)
    if num_steps < 1:
        num_steps = int(f32(chirp_t) * f32(rate))                      # :110
    length = int((f32(chirp_t) * f32(rate)) / f32(num_steps))          # :117
    if length < 1:
        length = 1                                                     # :120
        num_steps = int(f32(chirp_t) * f32(rate))                      # :121
    two32m1 = float(2 ** 32 - 1)
    chirpness_d = (two32m1 * (int(chirp_f) - int(freq0)) / (float(num_steps) - 1.0)) / float(rate)   # :125
    chirpness = int(chirpness_d) & 0xFFFFFFFF
    f0 = int(two32m1 * (float(freq0) / float(rate)))                   # :129
    return num_steps, length, chirpness, f0


def chirp_demod(x: np.ndarray, last_index: int, num_steps: int, length: int, chirpness: int, f0: int) -> np.ndarray:
    """ref: chirp_demodulator, cpp/kernels.cu:389-427; all index arithmetic in wrapping
    uint64 (`unsigned long`), then truncated to `int`."""
    x = np.asarray(x, dtype=np.complex64)
    n = x.shape[0]
    u64 = np.uint64
    with np.errstate(over="ignore"):
        eff = (u64(last_index) + np.arange(n, dtype=u64)) % u64(num_steps * length)       # :407
        fi = eff // u64(length)                                                            # :411
        q_phase = (fi // u64(2)) * (fi + u64(1)) + (fi % u64(2)) * ((fi + u64(1)) // u64(2))  # :413
        phase_corr = u64(chirpness) * (u64(length) * q_phase)                              # :416
        f0_u = u64(f0 & 0xFFFFFFFFFFFFFFFF)            # int -> unsigned long: sign extended
        idx64 = eff * (f0_u + fi * u64(chirpness)) - phase_corr                            # :419
    index = (idx64 & u64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)                     # (int)
    ph = index.astype(np.float64) / 2147483647.5                                           # :421
    cx = np.sin(np.pi * ph).astype(np.float32)         # chirp.x = sinpi(..)  (double -> float member)
    cy = (-np.cos(np.pi * ph)).astype(np.float32)      # chirp.y = -cospi(..)
    out = np.empty(n, dtype=np.complex64)
    out.real = cx * x.real + cy * x.imag                                                   # :424
    out.imag = cx * x.imag - cy * x.real                                                   # :425
    return out


class VnaHelper:
    """ref: class VNA_decimator_helper, cpp/USRP_server_memory_management.cpp:30-56."""

    def __init__(self, ppt: int, buffer_len: int):
        self.ppt, self.buffer_len = ppt, buffer_len
        self.total_len = buffer_len
        self.valid_size = self.total_len // ppt
        self.new0 = self.total_len - ppt * self.valid_size
        self.spare_begin = self.total_len - self.new0

    def update(self) -> None:
        self.total_len = self.buffer_len + self.new0
        self.valid_size = self.total_len // self.ppt
        self.new0 = self.total_len - self.ppt * self.valid_size
        self.spare_begin = self.total_len - self.new0


class Chirp:
    """ref: process_chirp, cpp/USRP_demodulator.cpp:342-397; lock-in = cublasCgemv(OP_T)
    with lda = ppt (cpp/kernels.cu:852-872; note the swapped argument names at the call,
    USRP_demodulator.cpp:363): S[v] = sum_p output[v*ppt + p] * profile[p]."""

    def __init__(self, rate: int, freq0: int, chirp_f: int, swipe_s: int, chirp_t: float,
                 decim: int, buffer_len: int, acc=np.complex64):
        self.ns, self.length, self.chirpness, self.f0 = chirp_params(rate, freq0, chirp_f, swipe_s, chirp_t)
        self.L, self.decim, self.acc = int(buffer_len), int(decim), acc
        self.last_index = 0                                            # :217
        self.spare_size = 0
        if self.decim > 0:
            self.ppt = self.length * self.decim                        # :231
            self.vna = VnaHelper(self.ppt, self.L)                     # :235
            self.profile = make_flat_window(self.ppt, self.ppt // 10)  # :246
            self.output = np.zeros(3 * self.L, dtype=np.complex64)     # :225

    def process(self, x: np.ndarray) -> np.ndarray:
        dem = chirp_demod(x, self.last_index, self.ns, self.length, self.chirpness, self.f0)   # :352
        self.last_index = (self.last_index + self.L) % (self.ns * self.length)                  # :355
        if self.decim <= 0:
            return dem                                                                           # :388-390
        self.output[self.spare_size: self.spare_size + self.L] = dem
        valid = self.vna.valid_size                                                              # :361
        A = self.output[: valid * self.ppt].reshape(valid, self.ppt).astype(self.acc, copy=False)
        res = (A @ self.profile.astype(self.acc)).astype(np.complex64)                           # :363
        self.spare_size = self.vna.new0                                                          # :369
        if self.spare_size > 0:                                                                  # :373-380
            sb = self.vna.spare_begin
            self.output[: self.vna.new0] = self.output[sb: sb + self.vna.new0].copy()
        self.vna.update()                                                                        # :382
        return res


# --------------------------------------------------------------------------
# TX tone comb (row f3)
# --------------------------------------------------------------------------
def tone_gen(freq, ampl, rate: int, scale: float = 1.0) -> np.ndarray:
    """ref: tone_gen, cpp/kernels.cu:589-684, literally: a zeroed vector of `rate` bins, the
    tones ASSIGNED to it (index = f if f > 0 else rate + f; later tones overwrite earlier ones),
    an UNNORMALISED inverse FFT of length `rate` (numpy's ifft divides by n: multiply back),
    an optional scale.  Returns the whole periodic buffer (use a small `rate`).

    A tone whose index falls outside the vector -- 0 Hz gives index == rate -- is a write
    past the allocation in the reference (undefined behaviour there); the transform never
    sees it, so it is left out here."""
    base = np.zeros(int(rate), dtype=np.complex64)                     # :611-614
    for f, a in zip(freq, ampl):                                       # :617-635
        idx = int(f) if int(f) > 0 else int(rate) + int(f)             # :622-624
        if 0 <= idx < rate:
            base[idx] = np.float32(a)                                  # :628, imaginary part 0 (:631)
    buf = np.fft.ifft(base.astype(np.complex128)) * float(rate)        # :641 CUFFT_INVERSE is unnormalised
    if scale != 1.0:
        buf = buf * np.float32(scale)                                  # :651
    return buf.astype(np.complex64)
