// ddc_kernels.hip -- fused per-tone digital down-converter for gfx950.
//
// One kernel serves both DIRECT (ref: direct_demodulator_integer
// cpp/kernels.cu:45-86 + FIR cpp/fir.cu:44-88 + cublasCgeam transpose
// cpp/USRP_demodulator.cpp:422-433) and TONES (ref: polyphase_filter
// cpp/kernels.cu:474-516 + cufftExecC2C cpp/USRP_demodulator.cpp:501 +
// tone_select cpp/kernels.cu:531-554), because both compute
//
//     y[n, G] = sum_{t < M*F} h[t] * x[(G-F+1)*M + t] * w_n^{s(t)}
//
// with w_n = exp(-2*pi*i*f_n/rate) (DIRECT: f_n in Hz, rate = sample rate;
// TONES: f_n = FFT bin, rate = nfft, M = nfft, F = pf_average), i.e. the
// reference's N*L intermediate, its N*(F+4) cuBLAS launches per buffer and its
// full-width FFT never exist here.
//
// Mapping (MI355X, wave64):
//   * one LANE per tone, one WAVE per (64 tones x chunk of consecutive input
//     blocks).  The IQ samples and the FIR taps are wave-uniform, so they are
//     fetched through the scalar data path (s_load -> SGPR) and enter the VALU
//     as scalar operands: no LDS traffic, no per-lane loads in the inner loop
//     (an LDS broadcast of the same bytes would cost 4x the LDS bandwidth the
//     CU has).
//   * NCO without per-sample sincos: the phasor of sample s = s0 + K*q + lo is
//     P_q * B[lo] where B[lo] = w_n^lo (lo < K) sits in VGPRs for the whole
//     kernel and P_q is advanced once per K samples in double precision from
//     an exact integer phase at the chunk start.  The inner loop is then
//     4 (mix) + 2F (real-tap MAC) FMA-class ops per tone-sample and the P_q
//     rotation is amortised over K samples.
//   * chunk boundaries: a chunk's first F-1 outputs miss the blocks of the
//     previous chunk; each chunk therefore writes its partial head sums to
//     `out`, its F-1 partial tail sums to `tails`, and ddc_fixup adds the two.
//     The tail of the last chunk is the stream carry for the next call (the
//     reference's FIR::_dout carry, cpp/fir.cu:64-69).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ddc_device.h"
#include "ddc_kernels.h"

namespace gsdr {

template <int F, int K>
__global__ __launch_bounds__(256) GSDR_NO_PK void ddc_kernel(
    const float2 *__restrict__ x,       // input samples, block b starts at x[b*M]
    const float *__restrict__ taps_t,   // [M][F]: taps_t[m*F+j] = h[j*M+m]
    const float2 *__restrict__ btab,    // [K][Npad]: w_n^lo
    const double2 *__restrict__ wk,     // [Npad]: w_n^K
    const double2 *__restrict__ wrem,   // [Npad]: w_n^(M mod K)
    const unsigned *__restrict__ fmod,  // [Npad]: f_n mod rate
    float2 *__restrict__ out,           // [rows][N] sample-major
    float2 *__restrict__ tails,         // [nch][F-1][Npad], slot c = tail of chunk c-1
    float2 *__restrict__ carry_out,     // [F-1][Npad] tail of the last chunk, or null
    DdcShape sh) {
    const int lane = threadIdx.x & 63;
    // wave-uniform ids must be SGPRs so that x/taps become scalar loads
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave = (int)blockIdx.x * 4 + wid;
    if (wave >= sh.TW * sh.nch) return;
    const int tw = wave % sh.TW;
    const int chunk = wave / sh.TW;
    const int n = tw * 64 + lane;
    const int M = sh.M, N = sh.N, Npad = sh.Npad;

    float2 B[K];
#pragma unroll
    for (int lo = 0; lo < K; ++lo) B[lo] = btab[(size_t)lo * Npad + n];
    const double2 WK = wk[n];
    const double2 WR = wrem[n];

    const int b0 = chunk_begin(chunk, sh);
    const int b1 = chunk_begin(chunk + 1, sh);

    // exact NCO phase at the first sample of the chunk
    // ref: kernels.cu:66-69  ii=(j+idx)%rate; phase=(tf*ii)%rate
    const unsigned long long s0 =
        mod_rate(sh.idx0 + (unsigned long long)b0 * sh.m_mod_rate, sh.rate, sh.rate_magic);
    const unsigned long long ph = mod_rate((unsigned long long)fmod[n] * s0, sh.rate, sh.rate_magic);
    double Pr, Pi;
    exact_phasor(ph, sh.inv_rate, Pr, Pi);

    float2 A[F];  // A[k]: partial sum of output G = b + k while block b is processed
#pragma unroll
    for (int k = 0; k < F; ++k) { A[k].x = 0.f; A[k].y = 0.f; }

    const int nfull = M / K;
    const int R = M - nfull * K;

    for (int b = b0; b < b1; ++b) {
        const float2 *__restrict__ xb = x + (size_t)b * M;
        int m = 0;
        for (int q = 0; q < nfull; ++q, m += K) {
            float2 S[F];
#pragma unroll
            for (int j = 0; j < F; ++j) { S[j].x = 0.f; S[j].y = 0.f; }
#pragma unroll
            for (int lo = 0; lo < K; ++lo) {
                const float2 xs = xb[m + lo];  // wave-uniform -> SGPR pair
                const float ur = xs.x * B[lo].x - xs.y * B[lo].y;
                const float ui = xs.x * B[lo].y + xs.y * B[lo].x;
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const float h = taps_t[(size_t)(m + lo) * F + j];  // wave-uniform
                    S[j].x = fmaf(h, ur, S[j].x);
                    S[j].y = fmaf(h, ui, S[j].y);
                }
            }
            const float pr = (float)Pr, pi = (float)Pi;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                // tap phase j of block b feeds output G = b + F-1-j  (fir.cu:56-61)
                A[F - 1 - j].x += pr * S[j].x - pi * S[j].y;
                A[F - 1 - j].y += pr * S[j].y + pi * S[j].x;
            }
            const double t = Pr * WK.x - Pi * WK.y;
            Pi = Pr * WK.y + Pi * WK.x;
            Pr = t;
        }
        if (R) {
            float2 S[F];
#pragma unroll
            for (int j = 0; j < F; ++j) { S[j].x = 0.f; S[j].y = 0.f; }
#pragma unroll
            for (int lo = 0; lo < K; ++lo) {
                if (lo < R) {
                    const float2 xs = xb[m + lo];
                    const float ur = xs.x * B[lo].x - xs.y * B[lo].y;
                    const float ui = xs.x * B[lo].y + xs.y * B[lo].x;
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        const float h = taps_t[(size_t)(m + lo) * F + j];
                        S[j].x = fmaf(h, ur, S[j].x);
                        S[j].y = fmaf(h, ui, S[j].y);
                    }
                }
            }
            const float pr = (float)Pr, pi = (float)Pi;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                A[F - 1 - j].x += pr * S[j].x - pi * S[j].y;
                A[F - 1 - j].y += pr * S[j].y + pi * S[j].x;
            }
            const double t = Pr * WR.x - Pi * WR.y;
            Pi = Pr * WR.y + Pi * WR.x;
            Pr = t;
        }
        // output G = b has now seen every block this chunk can give it
        if (b >= sh.g_off && n < N) out[(size_t)(b - sh.g_off) * N + n] = A[0];
#pragma unroll
        for (int k = 0; k + 1 < F; ++k) A[k] = A[k + 1];
        A[F - 1].x = 0.f;
        A[F - 1].y = 0.f;
    }

    if (F > 1) {
        float2 *dst = (chunk == sh.nch - 1) ? carry_out
                                            : tails + (size_t)(chunk + 1) * (F - 1) * Npad;
        if (dst) {
#pragma unroll
            for (int k = 0; k + 1 < F; ++k) dst[(size_t)k * Npad + n] = A[k];
        }
    }
}

// Sum over the 64 lanes of a wave with DPP row operations; the total arrives in lane 63 (see chirp_kernels.hip).
__device__ __forceinline__ float ddc_wave_sum63(float v) {
    int x = __float_as_int(v);
#define GSDR_DPP_ADD(ctrl, row_mask)                                                                    \
    x = __float_as_int(__int_as_float(x) +                                                              \
                       __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, 0xf, false)))
    GSDR_DPP_ADD(0xB1, 0xf);    // quad_perm [1,0,3,2]
    GSDR_DPP_ADD(0x4E, 0xf);    // quad_perm [2,3,0,1]
    GSDR_DPP_ADD(0x141, 0xf);   // row_half_mirror
    GSDR_DPP_ADD(0x140, 0xf);   // row_mirror: every lane holds the sum of its row of 16
    GSDR_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    GSDR_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef GSDR_DPP_ADD
    return __int_as_float(x);
}

// A handful of tones at a long decimation (round 3).  Every other DDC kernel gives a tone to a lane (or to a matrix
// column) and walks the samples of a block in sequence: with 16 tones at decim 1000 a launch is as long as ONE
// wave's (one workgroup's) walk through its blocks -- 72 us per 1 M-sample buffer on the matrix cores, 150 on the
// packed-FP32 kernel, for 0.35 GFLOP (profiles/r03_shape_sweep.log).  Here a WAVE is a (chunk, tone) pair and its 64
// lanes split the samples of a block: lane l takes t = l, l + 64, ...; its phasor is exact at t = l of every block
// (integer phase law, ref kernels.cu:66-69; sincos in double) and advances by w_n^64 in double; the F tap-phase sums
// of a block (ref fir.cu:48-61) are reduced over the wave with DPP adds.  Same chunk protocol as ddc_kernel: partial
// heads into `out`, F - 1 partial tails into `tails`, ddc_fixup adds them, the last chunk's tail is the carry.
template <int F>
__global__ __launch_bounds__(256) GSDR_NO_PK void ddc_few_kernel(
    const float2 *__restrict__ x, const float *__restrict__ taps_t, const unsigned *__restrict__ fmod,
    float2 *__restrict__ out, float2 *__restrict__ tails, float2 *__restrict__ carry_out, DdcShape sh) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave = (int)blockIdx.x * 4 + wid;
    const int M = sh.M, N = sh.N, Npad = sh.Npad;
    if (wave >= N * sh.nch) return;
    const int chunk = wave / N, n = wave - chunk * N;          // wave-uniform
    const unsigned f = fmod[n];
    double Wr, Wi;
    exact_phasor(mod_rate((unsigned long long)f * 64ull, sh.rate, sh.rate_magic), sh.inv_rate, Wr, Wi);
    const int b0 = chunk_begin(chunk, sh);
    const int b1 = chunk_begin(chunk + 1, sh);
    float2 A[F];
#pragma unroll
    for (int k = 0; k < F; ++k) { A[k].x = 0.f; A[k].y = 0.f; }
    for (int b = b0; b < b1; ++b) {
        const float2 *__restrict__ xb = x + (size_t)b * M;
        // ref: kernels.cu:66-69  ii=(j+idx)%rate; phase=(tf*ii)%rate, at this lane's first sample of the block
        const unsigned long long s0 =
            mod_rate(sh.idx0 + (unsigned long long)b * sh.m_mod_rate + (unsigned long long)lane, sh.rate, sh.rate_magic);
        double Pr, Pi;
        exact_phasor(mod_rate((unsigned long long)f * s0, sh.rate, sh.rate_magic), sh.inv_rate, Pr, Pi);
        float2 S[F];
#pragma unroll
        for (int j = 0; j < F; ++j) { S[j].x = 0.f; S[j].y = 0.f; }
        // U steps at a time, their loads first (clamped, not branched on): with the load inside the step that uses it a
        // wave paid a memory round trip per 64 samples -- 48 in a row at decim 1000, 31 us per buffer for ONE tone
        constexpr int U = 8;
        for (int t0 = lane; t0 < M; t0 += 64 * U) {
            float2 v[U];
            float h[U][F];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + 64 * u, tc = t < M ? t : M - 1;
                v[u] = xb[tc];
                const float *__restrict__ hp = taps_t + (size_t)tc * F;
#pragma unroll
                for (int j = 0; j < F; ++j) h[u][j] = hp[j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + 64 * u < M) {
                    const float pr = (float)Pr, pi = (float)Pi;
                    const float ur = v[u].x * pr - v[u].y * pi, ui = v[u].x * pi + v[u].y * pr;
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        S[j].x = fmaf(h[u][j], ur, S[j].x);
                        S[j].y = fmaf(h[u][j], ui, S[j].y);
                    }
                }
                const double tt = Pr * Wr - Pi * Wi;
                Pi = Pr * Wi + Pi * Wr;
                Pr = tt;
            }
        }
#pragma unroll
        for (int j = 0; j < F; ++j) {
            // tap phase j of block b feeds output G = b + F-1-j  (fir.cu:56-61); the sum of the wave is in lane 63
            A[F - 1 - j].x += ddc_wave_sum63(S[j].x);
            A[F - 1 - j].y += ddc_wave_sum63(S[j].y);
        }
        if (b >= sh.g_off && lane == 63) out[(size_t)(b - sh.g_off) * N + n] = A[0];
#pragma unroll
        for (int k = 0; k + 1 < F; ++k) A[k] = A[k + 1];
        A[F - 1].x = 0.f;
        A[F - 1].y = 0.f;
    }
    if (F > 1 && lane == 63) {
        float2 *dst = (chunk == sh.nch - 1) ? carry_out : tails + (size_t)(chunk + 1) * (F - 1) * Npad;
        if (dst) {
#pragma unroll
            for (int k = 0; k + 1 < F; ++k) dst[(size_t)k * Npad + n] = A[k];
        }
    }
}

// Adds the tail partial sums of chunk c-1 (or the stream carry, c == 0) to the
// head outputs of chunk c.  Tiny: nch*(F-1)*Npad threads.
__global__ GSDR_NO_PK void ddc_fixup(float2 *__restrict__ out, const float2 *__restrict__ tails,
                          const float2 *__restrict__ carry_in, float2 *__restrict__ carry_out,
                          int F, DdcShape sh) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Fm1 = F - 1;
    const long long total = (long long)sh.nch * Fm1 * sh.Npad;
    if (t >= total) return;
    const int n = (int)(t % sh.Npad);
    const int k = (int)((t / sh.Npad) % Fm1);
    const int c = (int)(t / ((long long)sh.Npad * Fm1));
    float2 src;
    if (c == 0) {
        if (!carry_in) return;
        src = carry_in[(size_t)k * sh.Npad + n];
    } else {
        src = tails[((size_t)c * Fm1 + k) * sh.Npad + n];
    }
    const int b0 = chunk_begin(c, sh);
    const int b1 = chunk_begin(c + 1, sh);
    const int G = b0 + k;
    if (G < b1) {
        if (G >= sh.g_off && n < sh.N) {
            float2 *o = out + (size_t)(G - sh.g_off) * sh.N + n;
            o->x += src.x;
            o->y += src.y;
        }
    } else if (carry_out) {
        // only reachable with a single chunk shorter than F-1 blocks (the host
        // guarantees nch == 1 then): the old carry outlives this buffer.
        float2 *o = carry_out + (size_t)(G - b1) * sh.Npad + n;
        o->x += src.x;
        o->y += src.y;
    }
}

// DIRECT with decim == 0: out[j][n] = x[j] * w_n^{idx+j}
// ref: direct_demodulator_integer (kernels.cu:45-86) + cublasCgeam transpose
// (USRP_demodulator.cpp:444-455).  HBM-write bound (8 B per tone-sample).
template <int K>
__global__ __launch_bounds__(256) GSDR_NO_PK void mix_kernel(
    const float2 *__restrict__ x, const float2 *__restrict__ btab,
    const double2 *__restrict__ wk, const unsigned *__restrict__ fmod,
    float2 *__restrict__ out, DdcShape sh) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave = (int)blockIdx.x * 4 + wid;
    if (wave >= sh.TW * sh.nch) return;
    const int tw = wave % sh.TW;
    const int chunk = wave / sh.TW;
    const int n = tw * 64 + lane;
    const int N = sh.N, Npad = sh.Npad;

    float2 B[K];
#pragma unroll
    for (int lo = 0; lo < K; ++lo) B[lo] = btab[(size_t)lo * Npad + n];
    const double2 WK = wk[n];

    // chunks are whole multiples of K samples (sh.M = samples per unit = K)
    const long long u0 = chunk_begin(chunk, sh);
    const long long u1 = chunk_begin(chunk + 1, sh);
    const long long total = sh.total;  // L
    const unsigned long long s0 =
        mod_rate(sh.idx0 + (unsigned long long)u0 * sh.m_mod_rate, sh.rate, sh.rate_magic);
    const unsigned long long ph = mod_rate((unsigned long long)fmod[n] * s0, sh.rate, sh.rate_magic);
    double Pr, Pi;
    exact_phasor(ph, sh.inv_rate, Pr, Pi);

    for (long long u = u0; u < u1; ++u) {
        const long long base = u * K;
        const float pr = (float)Pr, pi = (float)Pi;
#pragma unroll
        for (int lo = 0; lo < K; ++lo) {
            if (base + lo < total) {
                const float2 xs = x[base + lo];
                const float ur = xs.x * B[lo].x - xs.y * B[lo].y;
                const float ui = xs.x * B[lo].y + xs.y * B[lo].x;
                float2 o;
                o.x = pr * ur - pi * ui;
                o.y = pr * ui + pi * ur;
                if (n < N) out[(size_t)(base + lo) * N + n] = o;
            }
        }
        const double t = Pr * WK.x - Pi * WK.y;
        Pi = Pr * WK.y + Pi * WK.x;
        Pr = t;
    }
}

// The same for at most 32 tones: with a lane per tone a wave of mix_kernel has 64 - N idle lanes and
// writes N * 8 bytes per store (8 tones: 42 us per 1 M-sample buffer, 0.21 of the HBM peak).  Here a
// wave takes T = 2^tshift >= N tones x S = 64 / T consecutive samples at a time: lane = (sample phase
// sub, tone n), the lane's samples of a K-sample unit are lo = sub, sub + S, ...; one store covers
// S consecutive samples x T tones -- 512 contiguous bytes when T == N.  Same tables, same arithmetic per
// element (bit-identical to mix_kernel).
template <int K>
__global__ __launch_bounds__(256) GSDR_NO_PK void mix_small_kernel(
    const float2 *__restrict__ x, const float2 *__restrict__ btab,
    const double2 *__restrict__ wk, const unsigned *__restrict__ fmod,
    float2 *__restrict__ out, DdcShape sh, int tshift) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunk = (int)blockIdx.x * 4 + wid;
    if (chunk >= sh.nch) return;
    const int S = 64 >> tshift;                  // sample phases per wave, S <= K
    const int n = lane & ((1 << tshift) - 1), sub = lane >> tshift;
    const int N = sh.N, Npad = sh.Npad;

    float2 B[K / 2];                             // S >= 2: at most K / 2 samples per lane and unit
#pragma unroll
    for (int m = 0; m < K / 2; ++m) {
        const int lo = sub + S * m;
        B[m] = btab[(size_t)(lo < K ? lo : 0) * Npad + n];
    }
    const double2 WK = wk[n];
    const long long u0 = chunk_begin(chunk, sh);
    const long long u1 = chunk_begin(chunk + 1, sh);
    const long long total = sh.total;  // L
    const unsigned long long s0 =
        mod_rate(sh.idx0 + (unsigned long long)u0 * sh.m_mod_rate, sh.rate, sh.rate_magic);
    const unsigned long long ph = mod_rate((unsigned long long)fmod[n] * s0, sh.rate, sh.rate_magic);
    double Pr, Pi;
    exact_phasor(ph, sh.inv_rate, Pr, Pi);

    for (long long u = u0; u < u1; ++u) {
        const long long base = u * K;
        const float pr = (float)Pr, pi = (float)Pi;
#pragma unroll
        for (int m = 0; m < K / 2; ++m) {
            const int lo = sub + S * m;
            if (lo < K && base + lo < total) {
                const float2 xs = x[base + lo];
                const float ur = xs.x * B[m].x - xs.y * B[m].y;
                const float ui = xs.x * B[m].y + xs.y * B[m].x;
                float2 o;
                o.x = pr * ur - pi * ui;
                o.y = pr * ui + pi * ur;
                if (n < N) out[(size_t)(base + lo) * N + n] = o;
            }
        }
        const double t = Pr * WK.x - Pi * WK.y;
        Pi = Pr * WK.y + Pi * WK.x;
        Pr = t;
    }
}

// The same for at most 32 tones (round 3).  mix_small_kernel keeps mix_kernel's unit structure -- K samples per step of
// the block phasor, a table of K fp32 phasors inside the unit -- and with T = 1, 2 tones that leaves most lanes of
// a wave without a sample and every sample with ~30 instructions of 64-bit index arithmetic: 14 us per 1 M-sample
// buffer for one tone, whose 16 MB of traffic are 4 us of HBM time (profiles/r03_mix_rate.log).  Here a lane IS a
// (sample phase s, tone n) pair, the wave walks S = 64 / T consecutive samples per step, and each lane carries its
// own phasor: exact at its first sample (integer phase law, ref kernels.cu:66-69, sincos in double), advanced by
// w_n^S in double from step to step (at most a few dozen steps per wave), rounded to float for the one complex
// product per output (the reference: double sincospi, float product, kernels.cu:72-83).  One store covers S samples
// x T tones: 512 contiguous bytes when T == N.
__global__ __launch_bounds__(256) GSDR_NO_PK void mix_few_kernel(const float2 *__restrict__ x, const unsigned *__restrict__ fmod,
                                                                 float2 *__restrict__ out, DdcShape sh, int tshift, int steps) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int S = 64 >> tshift;
    const int n = lane & ((1 << tshift) - 1), s = lane >> tshift;
    const long long total = sh.total;
    const long long j0 = (long long)wave * steps * S + s;      // this lane's first sample
    if ((long long)wave * steps * S >= total) return;
    const unsigned f = fmod[n < sh.N ? n : 0];
    const unsigned long long s0 = mod_rate(sh.idx0 + (unsigned long long)j0, sh.rate, sh.rate_magic);
    double Pr, Pi, Wr, Wi;
    exact_phasor(mod_rate((unsigned long long)f * s0, sh.rate, sh.rate_magic), sh.inv_rate, Pr, Pi);
    exact_phasor(mod_rate((unsigned long long)f * (unsigned long long)S, sh.rate, sh.rate_magic), sh.inv_rate, Wr, Wi);
    const float2 *xp = x + j0;
    float2 *op = out + (size_t)j0 * sh.N + n;
    const size_t ostep = (size_t)S * sh.N;
    const bool tone = n < sh.N;
    long long j = j0;
#pragma unroll 4
    for (int i = 0; i < steps; ++i) {
        if (j < total) {
            const float2 v = *xp;
            const float pr = (float)Pr, pi = (float)Pi;
            float2 o;
            o.x = v.x * pr - v.y * pi;
            o.y = v.x * pi + v.y * pr;
            if (tone) *op = o;
        }
        const double t = Pr * Wr - Pi * Wi;
        Pi = Pr * Wi + Pi * Wr;
        Pr = t;
        j += S;
        xp += S;
        op += ostep;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int F, int K>
static hipError_t launch_ddc_fk(const DdcLaunch &a, hipStream_t st) {
    const int waves = a.sh.TW * a.sh.nch;
    const int grid = (waves + 3) / 4;
    hipLaunchKernelGGL((ddc_kernel<F, K>), dim3(grid), dim3(256), 0, st, a.x, a.taps_t, a.btab,
                       a.wk, a.wrem, a.fmod, a.out, a.tails, a.carry_out, a.sh);
    return hipGetLastError();
}

template <int K>
static hipError_t launch_ddc_k(int F, const DdcLaunch &a, hipStream_t st) {
    switch (F) {
        case 1: return launch_ddc_fk<1, K>(a, st);
        case 2: return launch_ddc_fk<2, K>(a, st);
        case 3: return launch_ddc_fk<3, K>(a, st);
        case 4: return launch_ddc_fk<4, K>(a, st);
        case 5: return launch_ddc_fk<5, K>(a, st);
        case 6: return launch_ddc_fk<6, K>(a, st);
        case 7: return launch_ddc_fk<7, K>(a, st);
        case 8: return launch_ddc_fk<8, K>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int F>
static hipError_t launch_ddc_few_f(const DdcLaunch &a, hipStream_t st) {
    const long long waves = (long long)a.sh.N * a.sh.nch;
    hipLaunchKernelGGL((ddc_few_kernel<F>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a.x, a.taps_t, a.fmod, a.out,
                       a.tails, a.carry_out, a.sh);
    return hipGetLastError();
}

static hipError_t launch_ddc_few(int F, const DdcLaunch &a, hipStream_t st) {
    if (a.sh.TW != 1) return hipErrorInvalidValue;          // (tones are waves here, not lanes: one table row)
    switch (F) {
        case 1: return launch_ddc_few_f<1>(a, st);
        case 2: return launch_ddc_few_f<2>(a, st);
        case 3: return launch_ddc_few_f<3>(a, st);
        case 4: return launch_ddc_few_f<4>(a, st);
        case 5: return launch_ddc_few_f<5>(a, st);
        case 6: return launch_ddc_few_f<6>(a, st);
        case 7: return launch_ddc_few_f<7>(a, st);
        case 8: return launch_ddc_few_f<8>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_ddc(int F, int K, const DdcLaunch &a, hipStream_t st, hipEvent_t stop) {
    // shapes are checked on the host: a kernel that writes past `tails` faults the GPU
    if (a.sh.nch < 1 || a.sh.nblk < 1 || a.sh.TW < 1 || (F > 1 && a.sh.nch > a.tails_nch) ||
        (F > 1 && a.sh.nch > 1 && a.sh.nblk / a.sh.nch < F - 1))
        return hipErrorInvalidValue;
    hipError_t e;
    if (a.few) e = launch_ddc_few(F, a, st);
    else if (a.pipe) e = launch_ddc_flat_main(F, K, a, st);
    else if (K == 16) e = launch_ddc_k<16>(F, a, st);
    else if (K == 32) e = launch_ddc_k<32>(F, a, st);
    else return hipErrorInvalidValue;
    if (e != hipSuccess) return e;
    if (stop) {
        e = hipEventRecord(stop, st);
        if (e != hipSuccess) return e;
    }
    if (F > 1) {
        const long long total = (long long)a.sh.nch * (F - 1) * a.sh.Npad;
        const int grid = (int)((total + 255) / 256);
        hipLaunchKernelGGL(ddc_fixup, dim3(grid), dim3(256), 0, st, a.out, a.tails, a.carry_in,
                           a.carry_out, F, a.sh);
        e = hipGetLastError();
    }
    return e;
}

// up to how many tones mix_few_kernel runs: same box, per 1 M-sample buffer, against mix_small_kernel / mix_kernel:
// 1 tone 6.8 against 14.3 us, 4: 11.4 / 15.8, 8: 16 / 21, 16: 27 / 32, 32: 68 / 77 - 82, 64: 144 / 110
// (profiles/r03_mix_rate.log).  GSDR_MIX_FEW=<n> moves the limit (0: the older kernels; read at every call, the
// tests use it).
static int mix_few_max() {
    const char *e = std::getenv("GSDR_MIX_FEW");
    return e && e[0] ? std::atoi(e) : 32;
}

hipError_t launch_mix(int K, const DdcLaunch &a, hipStream_t st) {
    if (a.sh.N <= mix_few_max() && a.sh.TW == 1 && a.sh.total > 0) {
        // very few tones: a lane per (sample, tone) with a phasor of its own (mix_few_kernel)
        int tshift = 0;
        while ((1 << tshift) < a.sh.N) ++tshift;
        const long long S = 64 >> tshift;
        const long long all_steps = (a.sh.total + S - 1) / S;
        long long steps = (all_steps + 8191) / 8192;            // ~8192 waves, at least 8 steps each
        if (steps < 8) steps = 8;
        const long long waves = (all_steps + steps - 1) / steps;
        hipLaunchKernelGGL(mix_few_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a.x, a.fmod, a.out, a.sh, tshift, (int)steps);
        return hipGetLastError();
    }
    if (a.sh.N <= 32 && a.sh.TW == 1 && (K == 16 || K == 32)) {
        // few tones: several sample phases per wave (mix_small_kernel); T >= 2 keeps S = 64 / T <= K / ... <= 32
        int tshift = 1;
        while ((1 << tshift) < a.sh.N) ++tshift;
        if ((64 >> tshift) > K) tshift = K == 16 ? 2 : 1;
        const int grid = (a.sh.nch + 3) / 4;
        if (K == 16)
            hipLaunchKernelGGL((mix_small_kernel<16>), dim3(grid), dim3(256), 0, st, a.x, a.btab, a.wk, a.fmod, a.out, a.sh, tshift);
        else
            hipLaunchKernelGGL((mix_small_kernel<32>), dim3(grid), dim3(256), 0, st, a.x, a.btab, a.wk, a.fmod, a.out, a.sh, tshift);
        return hipGetLastError();
    }
    const int waves = a.sh.TW * a.sh.nch;
    const int grid = (waves + 3) / 4;
    if (K == 16)
        hipLaunchKernelGGL((mix_kernel<16>), dim3(grid), dim3(256), 0, st, a.x, a.btab, a.wk,
                           a.fmod, a.out, a.sh);
    else if (K == 32)
        hipLaunchKernelGGL((mix_kernel<32>), dim3(grid), dim3(256), 0, st, a.x, a.btab, a.wk,
                           a.fmod, a.out, a.sh);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

const char *ddc_kernel_name() { return "ddc_kernel"; }
const char *ddc_few_kernel_name() { return "ddc_few_kernel"; }
const char *mix_kernel_name(int n_tones) { return n_tones <= mix_few_max() ? "mix_few_kernel" : n_tones <= 32 ? "mix_small_kernel" : "mix_kernel"; }

}  // namespace gsdr
