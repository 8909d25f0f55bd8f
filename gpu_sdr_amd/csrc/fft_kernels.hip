// fft_kernels.hip -- batched forward complex FFT of arbitrary length for the full-spectrum
// PFB (NOISE mode), hand-written for gfx950: replaces the reference's cuFFT plan
// (ref: cpp/USRP_demodulator.cpp:292-295 cufftPlanMany, :583 cufftExecC2C, relative to
// /root/reference) together with its polyphase_filter kernel (ref: cpp/kernels.cu:474-516).
//
// The work is HBM-bound integer-stride data movement with a few flops per byte, so the design
// is about passes over memory, not about matrix cores:
//   * Stockham autosort, decimation in time, one launch per radix stage: stage s with radix R
//     and p = product of the earlier radices reads x[i + r*(n/R)] (contiguous in i) and writes
//     y[(i - k)*R + k + r*p], k = i mod p -- no bit reversal, coalesced reads, every stage a
//     full read + write of the batch.  Radices 4 and 2 have their own butterflies, odd primes
//     up to 13 a generic O(R^2) one; twiddles come from one table w_n^k built in double on
//     the host (exact index: r*k*(n/(p*R)) < n).
//   * a length with a prime factor above 13 goes through Bluestein's chirp-z identity,
//         X[k] = conj(b_k) * sum_j (x_j conj(b_j)) b_(k-j),   b_j = exp(+i pi j^2 / n),
//     as a circular convolution of length m = 2^ceil(log2(2n-1)): two radix-4/2 FFTs of length
//     m per frame, the transform of the chirp computed once in double on the host.  j^2 mod 2n
//     is exact in 64-bit integers, so the chirp has no phase drift at any n.
// Error against an fp64 DFT: a few 1e-7 relative (tests/test_gpu_parity.py), bar 1e-5.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <vector>

#include "ddc_kernels.h"

namespace gsdr {

// No packed FP32 in these kernels: a NOISE handle may run beside the matrix-core DDC of another handle
// (two front-ends on one GPU), and v_pk_*_f32 with a high-half broadcast is unreliable in a wave that
// shares its SIMD with an MFMA loop (rule R3, DESIGN.md section 4.1, tools/ubench_pk_hazard.hip).

namespace {

// (make_float2 of the HIP headers is not always_inline: inside a kernel with other target features it
//  would stay a real call, s_swappc_b64 -- `make asm` + grep is the check)
__device__ __forceinline__ float2 mk2(float x, float y) {
    float2 v;
    v.x = x;
    v.y = y;
    return v;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return mk2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <int R>
__device__ __forceinline__ void butterfly(float2 (&u)[R], const float2 *__restrict__ tw, int n) {
    if constexpr (R == 2) {
        const float2 a = u[0], b = u[1];
        u[0] = mk2(a.x + b.x, a.y + b.y);
        u[1] = mk2(a.x - b.x, a.y - b.y);
    } else if constexpr (R == 4) {
        // forward DFT-4: w = -i
        const float2 a0 = mk2(u[0].x + u[2].x, u[0].y + u[2].y), a1 = mk2(u[0].x - u[2].x, u[0].y - u[2].y);
        const float2 b0 = mk2(u[1].x + u[3].x, u[1].y + u[3].y), b1 = mk2(u[1].x - u[3].x, u[1].y - u[3].y);
        u[0] = mk2(a0.x + b0.x, a0.y + b0.y);
        u[2] = mk2(a0.x - b0.x, a0.y - b0.y);
        u[1] = mk2(a1.x + b1.y, a1.y - b1.x);   // a1 - i*b1
        u[3] = mk2(a1.x - b1.y, a1.y + b1.x);   // a1 + i*b1
    } else {
        // odd prime: out[q] = sum_r u[r] * w_R^(q r), roots from the table (n is a multiple of R)
        float2 root[R];
        const int step = n / R;
#pragma unroll
        for (int m = 0; m < R; ++m) root[m] = tw[(size_t)m * step];
        float2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            float2 acc = u[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const float2 t = cmul(u[r], root[(q * r) % R]);
                acc.x += t.x;
                acc.y += t.y;
            }
            v[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = v[q];
    }
}

// one Stockham stage of radix R over `batch` transforms of length n; p = product of earlier radices
template <int R>
__global__ __launch_bounds__(256) GSDR_NO_PK void fft_pass_kernel(const float2 *__restrict__ x, float2 *__restrict__ y, int n, int p,
                                                       const float2 *__restrict__ tw, long long total) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int t = n / R;
    const long long b = g / t;
    const int i = (int)(g - b * t);
    const int k = i % p;
    const long long j = (long long)(i - k) * R + k;
    const float2 *xb = x + (size_t)b * n;
    float2 *yb = y + (size_t)b * n;
    const int tws = n / (p * R);
    float2 u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = xb[i + (size_t)r * t];
    if (k != 0) {
#pragma unroll
        for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[(size_t)r * k * tws]);
    }
    butterfly<R>(u, tw, n);
#pragma unroll
    for (int r = 0; r < R; ++r) yb[j + (size_t)r * p] = u[r];
}

// ref: polyphase_filter, cpp/kernels.cu:474-516 -- frames[r][k] = sum_i raw[(r+i)*n + k] * w[i*n + k],
// float accumulate in loop order, for the r < frames_n frames that are complete
__global__ __launch_bounds__(256) GSDR_NO_PK void pfb_filter_kernel(const float2 *__restrict__ raw, const float *__restrict__ w, int n,
                                                         int avg, long long total, float2 *__restrict__ frames) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int k = (int)(g % n);
    float2 acc = mk2(0.f, 0.f);
    for (int i = 0; i < avg; ++i) {
        const float2 s = raw[g + (size_t)i * n];
        const float wi = w[(size_t)i * n + k];
        acc.x += s.x * wi;
        acc.y += s.y * wi;
    }
    frames[g] = acc;
}

// Bluestein, step 1: a[b][j] = x[b][j] * conj(chirp[j]) for j < n, zero up to m
__global__ __launch_bounds__(256) GSDR_NO_PK void bluestein_pre_kernel(const float2 *__restrict__ x, const float2 *__restrict__ chirp, int n,
                                                            int m, long long total, float2 *__restrict__ a) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const long long b = g / m;
    const int j = (int)(g - b * m);
    float2 v = mk2(0.f, 0.f);
    if (j < n) {
        const float2 c = chirp[j];
        v = cmul(x[(size_t)b * n + j], mk2(c.x, -c.y));
    }
    a[g] = v;
}

// step 2: d = conj(A * Bhat): the inverse transform is then a forward one (IFFT(z) = conj(FFT(conj z))/m)
__global__ __launch_bounds__(256) GSDR_NO_PK void bluestein_mul_kernel(float2 *__restrict__ a, const float2 *__restrict__ bhat, int m, long long total) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const float2 v = cmul(a[g], bhat[g % m]);
    a[g] = mk2(v.x, -v.y);
}

// step 3: X[b][k] = conj(chirp[k]) * conj(e[b][k]) / m, k < n
__global__ __launch_bounds__(256) GSDR_NO_PK void bluestein_post_kernel(const float2 *__restrict__ e, const float2 *__restrict__ chirp, int n,
                                                             int m, long long total, float2 *__restrict__ out) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const long long b = g / n;
    const int k = (int)(g - b * n);
    const float2 c = chirp[k], v = e[(size_t)b * m + k];
    const float inv = 1.f / (float)m;
    const float2 r = cmul(mk2(c.x, -c.y), mk2(v.x, -v.y));
    out[g] = mk2(r.x * inv, r.y * inv);
}

// ---------------------------------------------------------------------------------------------
// The whole PFB of a frame in one workgroup (TONES and NOISE with fft_tones <= kPfbLdsMaxN):
// polyphase filter -> Stockham stages between two LDS buffers -> bin selection -> output.
// ref: process_pfb / process_pfb_spec, cpp/USRP_demodulator.cpp:486-565, :568-649 (polyphase_filter
// kernels.cu:474-516, cufftExecC2C, tone_select kernels.cu:520-554).  Per 1 Mi-sample buffer that is
// one read of the window (every sample F times, the repeats out of L2) and one write of the
// selected bins: HBM-bound, a few flops per byte -- no staging copy of the buffer, no pass through
// memory per radix stage.  The logical window is [carry | in]: the samples the previous call left
// over, then the new buffer; the last workgroups of the grid copy this call's leftovers to the
// carry of the next one.
// ---------------------------------------------------------------------------------------------
// Diagnostic builds only (-DGSDR_STAMP_BUILD, scratch/stamp_pfb.py): every workgroup leaves the times
// (s_memrealtime, 100 MHz) of its phases in a buffer of its own; the shipped library has none of this.
#ifdef GSDR_STAMP_BUILD
__device__ unsigned long long *g_fft_stamp_buf = nullptr;
__device__ __forceinline__ void fft_stamp(int slot) {
    if (g_fft_stamp_buf && threadIdx.x == 0) g_fft_stamp_buf[8 * (size_t)blockIdx.x + slot] = __builtin_amdgcn_s_memrealtime();
}
#else
__device__ __forceinline__ void fft_stamp(int) {}
#endif

struct PfbLdsArgs {
    const float2 *carry;       // new_0 samples left over by the previous call
    const float2 *in;          // the new buffer
    const float *window;       // [F][n] taps
    const float2 *tw;          // w_n^k, k < n
    const int *sel;            // selected bins (nullptr: all n bins)
    float2 *out;               // [frames_n][n_out]
    float2 *carry_out;         // receives W[spare_begin .. spare_begin + spare_n)
    int n, F, frames_n, n_out, new_0, FR;
    int spare_begin, spare_n;
    unsigned main_blocks, blocks_per_xcd;
    int n_radices;
    int radices[16];
    // x / d as umulhi(x, magic(d)) (0: d == 1), exact while x * d < 2^32 -- every quotient of the kernel
    // is of an index below 2^17 by a divisor of at most 2^13.  A runtime integer division is ~30 vector
    // instructions; with one or two per butterfly the kernel was bound by them.
    unsigned mag_n, mag_nout;
    unsigned mag_t[16], mag_p[16];     // per stage: t = n / R, p = product of the earlier radices
    int stage_t[16], stage_tws[16];    // per stage: t and n / (p R), so that no stage divides at run time
    unsigned mag_t4;                   // prime-first stage: t4 = ceil(t / 4)
};

__device__ __forceinline__ int fdiv(int x, unsigned magic) {
    return magic ? (int)__umulhi((unsigned)x, magic) : x;
}

__device__ __forceinline__ float2 pfb_window_at(const PfbLdsArgs &a, int q) {
    const float2 *p = q < a.new_0 ? a.carry + q : a.in + (q - a.new_0);
    return *p;
}

template <int R>
__device__ __forceinline__ void lds_stage(const float2 *src, float2 *dst, int n, int p, int t, int tws, unsigned mag_t,
                                          unsigned mag_p, const float2 *__restrict__ tw, int FR, int tid, int NT) {
    for (int g = tid; g < FR * t; g += NT) {
        const int fr = FR == 1 ? 0 : fdiv(g, mag_t), i = g - fr * t;
        const int k = i - fdiv(i, mag_p) * p;
        const int j = (i - k) * R + k;
        const float2 *xb = src + fr * n;
        float2 *yb = dst + fr * n;
        float2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = xb[i + r * t];
        if (p != 1) {                             // the first stage has no twiddles in front (k == 0)
            const int kt = k * tws;
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[r * kt]);
        }
        butterfly<R>(u, tw, n);
#pragma unroll
        for (int r = 0; r < R; ++r) yb[j + r * p] = u[r];
    }
}

// A larger prime radix R as the FIRST stage (p = 1: no twiddles in front of the butterfly),
//     out[i R + q] = sum_r x[i + r t] w_R^(q r),   t = n / R.
// The terms r and R - r are taken together, x_r w^(qr) + x_(R-r) w^(-qr) = S_r cos - i D_r sin with
// S_r = x_r + x_(R-r), D_r = x_r - x_(R-r) (formed once, in place, by a first pass), and the two sums
//     A = sum_(r <= h) S_r cos(2 pi q r / R),   B = sum_(r <= h) D_r sin(2 pi q r / R),   h = (R - 1) / 2,
// give two outputs, out[q] = x_0 + A - iB and out[R - q] = x_0 + A + iB: a quarter of the multiply-adds
// of the plain sum (which cost 11 of the 27 us of a 1230 = 41*2*3*5-point buffer).  One q and four
// consecutive columns i per work item; roots: w_R^m = (cos, -sin)(2 pi m / R), m < R, in the LDS.
__device__ __forceinline__ void lds_stage_prime_first(int R, float2 *src, float2 *dst, int n, const float2 *roots,
                                                      int t, unsigned mag_t, unsigned mag_t4, int FR, int tid, int NT) {
    const int t4 = (t + 3) >> 2, h = (R - 1) >> 1;
    // pass 1: S and D in place
    for (int g = tid; g < FR * h * t; g += NT) {
        const int rr = fdiv(g, mag_t), i = g - rr * t;          // rr = fr * h + (r - 1)
        const int fr = FR == 1 ? 0 : rr / h, r = rr - fr * h + 1;
        const int lo = fr * n + i + r * t, hi = fr * n + i + (R - r) * t;
        const float2 x = src[lo], y = src[hi];
        src[lo] = mk2(x.x + y.x, x.y + y.y);
        src[hi] = mk2(x.x - y.x, x.y - y.y);
    }
    __syncthreads();
    // pass 2: q = 0 .. h; work item = (frame, q, group of four columns)
    const int items = (h + 1) * t4;
    for (int g = tid; g < FR * items; g += NT) {
        const int fr = FR == 1 ? 0 : g / items, rem = g - fr * items;
        const int q = fdiv(rem, mag_t4), i0 = (rem - q * t4) << 2;
        int off[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) off[c] = fr * n + (i0 + c < t ? i0 + c : t - 1);   // lanes beyond t redo the last column
        float2 x0[4], A[4], B[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            x0[c] = src[off[c]];
            A[c] = mk2(0.f, 0.f);
            B[c] = mk2(0.f, 0.f);
        }
        int idx = 0;
        for (int r = 1; r <= h; ++r) {
            idx += q;
            idx = idx >= R ? idx - R : idx;
            const float2 w = roots[idx];             // (cos, -sin)
            const float cs = w.x, sn = -w.y;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float2 S = src[off[c] + r * t], D = src[off[c] + (R - r) * t];
                A[c].x = fmaf(S.x, cs, A[c].x);
                A[c].y = fmaf(S.y, cs, A[c].y);
                B[c].x = fmaf(D.x, sn, B[c].x);
                B[c].y = fmaf(D.y, sn, B[c].y);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (i0 + c < t) {
                const int o = fr * n + (i0 + c) * R;
                const float bx = x0[c].x + A[c].x, by = x0[c].y + A[c].y;
                dst[o + q] = mk2(bx + B[c].y, by - B[c].x);
                if (q != 0) dst[o + R - q] = mk2(bx - B[c].y, by + B[c].x);
            }
    }
}

// any further prime radix above 13 (two large prime factors in one length: rare): one output per work
// item, out[q] = sum_r x[i + r t] w_n^(r (k tws + q n/R)), roots from the twiddle table
__device__ __forceinline__ void lds_stage_generic(int R, const float2 *src, float2 *dst, int n, int p,
                                                  const float2 *__restrict__ tw, int FR, int tid, int NT) {
    const int t = n / R, tws = n / (p * R), nr = n / R;
    for (int g = tid; g < FR * t * R; g += NT) {
        const int fr = g / (t * R), rem = g - fr * (t * R);
        const int q = rem / t, i = rem - q * t;
        const int k = i % p;
        const int j = (i - k) * R + k;
        const float2 *xb = src + fr * n + i;
        const int e = (int)(((long long)k * tws + (long long)q * nr) % n);
        float2 acc = xb[0];
        int idx = 0;
        for (int r = 1; r < R; ++r) {
            idx += e;
            if (idx >= n) idx -= n;
            const float2 v = cmul(xb[r * t], tw[idx]);
            acc.x += v.x;
            acc.y += v.y;
        }
        dst[fr * n + j + q * p] = acc;
    }
}

// TWL: the twiddle table w_n^k is copied into the LDS first (frames of up to kPfbLdsTwMaxN points); the
// stages then find their factors at LDS latency instead of one L1 round trip per stage.
// NT: threads per workgroup -- 256, or 512 (1024 would cap a thread at 128 registers: spills) for frames of 2048 points and more (a buffer has few of them:
// the frame's own parallelism has to fill the compute unit)
template <bool TWL, int NT>
__global__ __launch_bounds__(NT) GSDR_NO_PK void pfb_lds_kernel(const PfbLdsArgs a) {
    extern __shared__ float2 pfb_lds[];
    const int tid = threadIdx.x, n = a.n, FR = a.FR;
    // Workgroups are dealt to the 8 XCDs in turn (blockIdx % 8) and each XCD has its own L2.  Frame r
    // shares (F-1)/F of its samples with frame r+1: give every XCD a contiguous run of frames, so that
    // the repeats come out of its L2 (with frame = blockIdx every sample was fetched F times over the
    // fabric: FETCH_SIZE 32 MB per 8 MB buffer).  The grid is padded to a multiple of 8.
    const unsigned padded = a.blocks_per_xcd * 8u;
    if (blockIdx.x >= padded) {
        // leftovers of this call -> carry of the next one
        const int j0 = (int)(blockIdx.x - padded) * 2048;
        for (int j = j0 + tid; j < j0 + 2048 && j < a.spare_n; j += NT)
            a.carry_out[j] = pfb_window_at(a, a.spare_begin + j);
        return;
    }
    fft_stamp(0);
    float2 *A = pfb_lds, *B = pfb_lds + (size_t)FR * n, *roots = pfb_lds + (size_t)2 * FR * n;
    float2 *twl = roots + (kPfbLdsMaxPrime + 1);
    const unsigned wg = (blockIdx.x & 7u) * a.blocks_per_xcd + (blockIdx.x >> 3);
    if (wg >= a.main_blocks) return;
    const int f0 = (int)wg * FR;
    // polyphase filter: float accumulate in tap order (as pfb_filter_kernel).  Four points per thread and
    // up to four taps at a time: their 16 + 16 loads are issued before the first product (a loop of
    // load -> multiply-add is one memory round trip per tap and point: 4 us of a 1024-point frame).
    // The twiddle table and the roots of a large first radix (w_R^m = w_n^(m n/R)) travel with the
    // first batch of loads.
    const int R0 = a.n_radices > 0 && a.radices[0] > 13 ? a.radices[0] : 0;
    // the parameters of stage s live in lane s of three registers
    const int st_lane = tid & 15;
    const int st_radix = a.radices[st_lane], st_mag_t = (int)a.mag_t[st_lane], st_mag_p = (int)a.mag_p[st_lane];
    const int st_t = a.stage_t[st_lane], st_tws = a.stage_tws[st_lane];
    // the bins of the first output columns of this thread: loaded now, used behind the last stage
    int sel0[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int idx = tid + NT * c;
        const int fr = FR == 1 ? 0 : fdiv(idx, a.mag_nout);
        const int u = idx - fr * a.n_out;
        sel0[c] = a.sel && idx < FR * a.n_out ? a.sel[u] : u;
    }
    bool first = true;
    const bool all_in = f0 * n >= a.new_0 && a.frames_n - f0 >= FR;     // uniform; q >= new_0 for every read then
    const float2 *in_base = a.in - a.new_0;
    for (int base = tid; base < FR * n; base += 4 * NT) {
        int kk[4], q0[4];
        bool ok[4];
        float2 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = base + NT * c;
            const int idc = idx < FR * n ? idx : FR * n - 1;
            const int fr = FR == 1 ? 0 : fdiv(idc, a.mag_n);
            kk[c] = idc - fr * n;
            ok[c] = idx < FR * n && f0 + fr < a.frames_n;
            q0[c] = ok[c] ? (f0 + fr) * n + kk[c] : a.new_0; // W[new_0] = in[0] is always there
            acc[c] = mk2(0.f, 0.f);
        }
        for (int i0 = 0; i0 < a.F; i0 += 4) {
            float2 sm[4][4];
            float wv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j < a.F ? i0 + j : a.F - 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int q = q0[c] + (ok[c] ? i * n : 0);
                    // (a workgroup whose frames lie behind the carried samples reads the buffer through one
                    //  uniform base and 32-bit offsets; the select of pfb_window_at() is 64-bit arithmetic per load)
                    sm[c][j] = all_in ? in_base[(unsigned)q] : pfb_window_at(a, q);
                    wv[c][j] = a.window[(unsigned)(i * n + kk[c])];
                }
            }
            if (first) {
                first = false;
                if (TWL)
                    for (int k = tid; k < n; k += NT) twl[k] = a.tw[k];
                for (int m = tid; m < R0; m += NT) roots[m] = a.tw[m * (n / R0)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < a.F) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c].x += sm[c][j].x * wv[c][j];
                        acc[c].y += sm[c][j].y * wv[c][j];
                    }
                }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (base + NT * c < FR * n) A[base + NT * c] = ok[c] ? acc[c] : mk2(0.f, 0.f);
    }
    __syncthreads();
    fft_stamp(1);
    const float2 *tw = TWL ? twl : a.tw;
    float2 *src = A, *dst = B;
    int p = 1;
    for (int s = 0; s < a.n_radices; ++s) {
        // (a.radices[s] with a running s is a scalar load and its latency in every stage)
        const int R = __builtin_amdgcn_readlane(st_radix, s);
        const unsigned mt = (unsigned)__builtin_amdgcn_readlane(st_mag_t, s), mp = (unsigned)__builtin_amdgcn_readlane(st_mag_p, s);
        const int st = __builtin_amdgcn_readlane(st_t, s), stw = __builtin_amdgcn_readlane(st_tws, s);
        switch (R) {
            case 2: lds_stage<2>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 3: lds_stage<3>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 4: lds_stage<4>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 5: lds_stage<5>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 7: lds_stage<7>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 11: lds_stage<11>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 13: lds_stage<13>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            default:
                if (s == 0)
                    lds_stage_prime_first(R, src, dst, n, roots, st, mt, a.mag_t4, FR, tid, NT);
                else
                    lds_stage_generic(R, src, dst, n, p, tw, FR, tid, NT);
                break;
        }
        __syncthreads();
        fft_stamp(2 + (s < 4 ? s : 4));
        p *= R;
        float2 *t2 = src;
        src = dst;
        dst = t2;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int idx = tid + NT * c;
        const int fr = FR == 1 ? 0 : fdiv(idx, a.mag_nout), u = idx - fr * a.n_out;
        if (idx < FR * a.n_out && f0 + fr < a.frames_n) a.out[(size_t)(f0 + fr) * a.n_out + u] = src[fr * n + sel0[c]];
    }
    for (int idx = tid + 4 * NT; idx < FR * a.n_out; idx += NT) {
        const int fr = FR == 1 ? 0 : fdiv(idx, a.mag_nout), u = idx - fr * a.n_out;
        const int r = f0 + fr;
        if (r < a.frames_n) a.out[(size_t)r * a.n_out + u] = src[fr * n + (a.sel ? a.sel[u] : u)];
    }
    fft_stamp(7);
}


inline unsigned grid_for(long long total) { return (unsigned)((total + 255) / 256); }

template <int R>
hipError_t launch_pass(const float2 *x, float2 *y, int n, int p, const float2 *tw, int batch, hipStream_t st) {
    const long long total = (long long)batch * (n / R);
    if (total <= 0 || total > 0x7fffffffLL * 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL((fft_pass_kernel<R>), dim3(grid_for(total)), dim3(256), 0, st, x, y, n, p, tw, total);
    return hipGetLastError();
}

hipError_t launch_radix(int R, const float2 *x, float2 *y, int n, int p, const float2 *tw, int batch, hipStream_t st) {
    switch (R) {
        case 2: return launch_pass<2>(x, y, n, p, tw, batch, st);
        case 3: return launch_pass<3>(x, y, n, p, tw, batch, st);
        case 4: return launch_pass<4>(x, y, n, p, tw, batch, st);
        case 5: return launch_pass<5>(x, y, n, p, tw, batch, st);
        case 7: return launch_pass<7>(x, y, n, p, tw, batch, st);
        case 11: return launch_pass<11>(x, y, n, p, tw, batch, st);
        case 13: return launch_pass<13>(x, y, n, p, tw, batch, st);
        default: return hipErrorInvalidValue;
    }
}

// radices of n (4s first, then 2, then odd primes up to 13); empty when a larger prime remains
std::vector<int> factorize(int n) {
    std::vector<int> r;
    int m = n;
    while (m % 4 == 0) { r.push_back(4); m /= 4; }
    for (int p : {2, 3, 5, 7, 11, 13})
        while (m % p == 0) { r.push_back(p); m /= p; }
    if (m != 1) r.clear();
    return r;
}

void host_twiddles(int len, std::vector<float2> &tw) {
    tw.resize((size_t)len);
    for (int k = 0; k < len; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)len;
        tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
}

// in-place radix-2 double FFT on the host (plan time only: the chirp's transform)
void host_fft_pow2(std::vector<double> &re, std::vector<double> &im) {
    const size_t m = re.size();
    for (size_t i = 1, j = 0; i < m; ++i) {
        size_t bit = m >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= m; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < m; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = std::cos(ang * (double)k), wi = std::sin(ang * (double)k);
                const size_t a = i + k, b = i + k + len / 2;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
}

template <typename T>
hipError_t to_device(T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    hipError_t e = hipMalloc((void **)dst, (src.empty() ? 1 : src.size()) * sizeof(T));
    if (e != hipSuccess) return e;
    return src.empty() ? hipSuccess : hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace

#ifdef GSDR_STAMP_BUILD
extern "C" int gsdr_debug_set_fft_stamp_buffer(void *dev_ptr) {
    unsigned long long *p = (unsigned long long *)dev_ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fft_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

int fft_plan_build(FftPlan &pl, int n) {
    pl = FftPlan{};
    if (n < 1) return -1;
    pl.n = n;
    std::vector<int> rad = factorize(n);
    std::vector<float2> tw;
    if (!rad.empty() || n == 1) {
        pl.m = 0;
        pl.n_radices = (int)rad.size();
        for (size_t i = 0; i < rad.size() && i < 32; ++i) pl.radices[i] = rad[i];
        host_twiddles(n, tw);
        if (to_device(&pl.d_tw, tw) != hipSuccess) return -1;
        return 0;
    }
    // Bluestein: m = 2^ceil(log2(2n-1))
    long long m = 1;
    while (m < 2LL * n - 1) m <<= 1;
    if (m > (1LL << 26)) return -1;
    pl.m = (int)m;
    rad = factorize(pl.m);
    pl.n_radices = (int)rad.size();
    for (size_t i = 0; i < rad.size() && i < 32; ++i) pl.radices[i] = rad[i];
    host_twiddles(pl.m, tw);
    if (to_device(&pl.d_tw, tw) != hipSuccess) return -1;
    std::vector<float2> chirp((size_t)n);
    std::vector<double> br((size_t)m, 0.0), bi((size_t)m, 0.0);
    for (long long j = 0; j < n; ++j) {
        const long long q = (j * j) % (2LL * n);          // exact: j < 2^26
        const double a = M_PI * (double)q / (double)n;
        const double c = std::cos(a), s = std::sin(a);
        chirp[(size_t)j] = make_float2((float)c, (float)s);
        br[(size_t)j] = c; bi[(size_t)j] = s;
        if (j) { br[(size_t)(m - j)] = c; bi[(size_t)(m - j)] = s; }
    }
    host_fft_pow2(br, bi);
    std::vector<float2> bhat((size_t)m);
    for (long long k = 0; k < m; ++k) bhat[(size_t)k] = make_float2((float)br[(size_t)k], (float)bi[(size_t)k]);
    if (to_device(&pl.d_chirp, chirp) != hipSuccess || to_device(&pl.d_bhat, bhat) != hipSuccess) return -1;
    return 0;
}

void fft_plan_free(FftPlan &pl) {
    if (pl.d_tw) (void)hipFree(pl.d_tw);
    if (pl.d_chirp) (void)hipFree(pl.d_chirp);
    if (pl.d_bhat) (void)hipFree(pl.d_bhat);
    pl = FftPlan{};
}

// Stockham stages of one length-`len` transform set: src -> ... -> dst, ping-ponging through tmp
// (src is destroyed).  With no stage at all (len == 1) the data is copied.
static hipError_t run_stages(const FftPlan &pl, int len, float2 *src, float2 *dst, float2 *tmp, int batch, hipStream_t st) {
    if (pl.n_radices == 0)
        return hipMemcpyAsync(dst, src, (size_t)batch * len * sizeof(float2), hipMemcpyDeviceToDevice, st);
    float2 *cur = src;
    int p = 1;
    for (int s = 0; s < pl.n_radices; ++s) {
        const bool last = s == pl.n_radices - 1;
        // destinations alternate so that the last stage lands in dst and no stage writes its own input
        float2 *to = last ? dst : (cur == tmp ? src : tmp);
        if (to == cur) return hipErrorInvalidValue;
        hipError_t e = launch_radix(pl.radices[s], cur, to, len, p, pl.d_tw, batch, st);
        if (e != hipSuccess) return e;
        p *= pl.radices[s];
        cur = to;
    }
    return hipSuccess;
}

hipError_t fft_forward(const FftPlan &pl, float2 *src, float2 *dst, float2 *tmp, int batch, hipStream_t st) {
    if (batch < 1 || !src || !dst || !tmp || dst == src || dst == tmp || src == tmp) return hipErrorInvalidValue;
    if (pl.m == 0) return run_stages(pl, pl.n, src, dst, tmp, batch, st);
    // Bluestein: src [batch][n] -> tmp [batch][m] (a) -> FFT_m -> src' ... the three scratch roles
    // rotate between `src` and `tmp`, both sized for batch*m; `dst` only receives the final [batch][n]
    const int n = pl.n, m = pl.m;
    const long long tot_m = (long long)batch * m, tot_n = (long long)batch * n;
    hipLaunchKernelGGL(bluestein_pre_kernel, dim3(grid_for(tot_m)), dim3(256), 0, st, src, pl.d_chirp, n, m, tot_m, tmp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // FFT_m: tmp -> src (through dst?  dst may be too small for batch*m: use the two scratch buffers only)
    // stages alternate tmp <-> src; force the last one into src by choosing the start accordingly
    {
        float2 *cur = tmp, *other = src;
        int p = 1;
        for (int s = 0; s < pl.n_radices; ++s) {
            e = launch_radix(pl.radices[s], cur, other, m, p, pl.d_tw, batch, st);
            if (e != hipSuccess) return e;
            p *= pl.radices[s];
            std::swap(cur, other);
        }
        // result in `cur`
        hipLaunchKernelGGL(bluestein_mul_kernel, dim3(grid_for(tot_m)), dim3(256), 0, st, cur, pl.d_bhat, m, tot_m);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        p = 1;
        for (int s = 0; s < pl.n_radices; ++s) {
            e = launch_radix(pl.radices[s], cur, other, m, p, pl.d_tw, batch, st);
            if (e != hipSuccess) return e;
            p *= pl.radices[s];
            std::swap(cur, other);
        }
        hipLaunchKernelGGL(bluestein_post_kernel, dim3(grid_for(tot_n)), dim3(256), 0, st, cur, pl.d_chirp, n, m, tot_n, dst);
        e = hipGetLastError();
    }
    return e;
}

// ref: tone_select, cpp/kernels.cu:520-554 -- out[frame][u] = spectra[frame][bin(u)]
__global__ __launch_bounds__(256) GSDR_NO_PK void pfb_select_kernel(const float2 *__restrict__ spectra, int n, const int *__restrict__ sel,
                                                         int n_out, long long total, float2 *__restrict__ out) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const long long r = g / n_out;
    const int u = (int)(g - r * n_out);
    out[g] = spectra[(size_t)r * n + sel[u]];
}

hipError_t launch_pfb_select(const float2 *spectra, int nfft, int frames_n, const int *sel, int n_out, float2 *out, hipStream_t st) {
    if (nfft < 1 || frames_n < 1 || n_out < 1 || !spectra || !sel || !out) return hipErrorInvalidValue;
    const long long total = (long long)frames_n * n_out;
    hipLaunchKernelGGL(pfb_select_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, spectra, nfft, sel, n_out, total, out);
    return hipGetLastError();
}

hipError_t launch_pfb_filter(const float2 *raw, const float *window, int nfft, int avg, int frames_n, float2 *frames,
                             hipStream_t st) {
    if (nfft < 1 || avg < 1 || frames_n < 1 || !raw || !window || !frames) return hipErrorInvalidValue;
    const long long total = (long long)frames_n * nfft;
    hipLaunchKernelGGL(pfb_filter_kernel, dim3(grid_for(total)), dim3(256), 0, st, raw, window, nfft, avg, total, frames);
    return hipGetLastError();
}

// Stages of the in-LDS transform: 4s, 2, then every odd prime factor (3, 5, 7 with register
// butterflies, any larger prime through the one-output-per-item stage).  Empty when n does not fit:
// n > kPfbLdsMaxN, more than 16 stages, or a prime factor above kPfbLdsMaxPrime (its stage is O(R) per
// output: a 1021-point prime frame would be a plain DFT).
int pfb_lds_plan(int n, int *radices) {
    if (n < 1 || n > kPfbLdsMaxN) return -1;
    int cnt = 0, m = n;
    auto push = [&](int r) { if (cnt < 16) radices[cnt] = r; ++cnt; };
    // prime factors above 13 first, the largest in front (its stage then needs no twiddles)
    int small = 1;
    for (int q : {2, 3, 5, 7, 11, 13})
        while (m % q == 0) { small *= q; m /= q; }
    int big[16], nbig = 0;
    for (int q = 17; q <= m; q += 2)
        while (m % q == 0) {
            if (q > kPfbLdsMaxPrime || nbig == 16) return -1;
            big[nbig++] = q;
            m /= q;
        }
    if (m != 1) return -1;
    for (int i = nbig - 1; i >= 0; --i) push(big[i]);
    m = small;
    while (m % 4 == 0) { push(4); m /= 4; }
    for (int q : {2, 3, 5, 7, 11, 13})
        while (m % q == 0) { push(q); m /= q; }
    return cnt <= 16 ? cnt : -1;
}

hipError_t launch_pfb_lds(const float2 *carry, int new_0, const float2 *in, const float *window, const float2 *tw,
                          int nfft, int avg, int frames_n, const int *sel, int n_out, float2 *out,
                          float2 *carry_out, int spare_begin, int spare_n, long long window_len, hipStream_t st) {
    PfbLdsArgs a{};
    a.n_radices = pfb_lds_plan(nfft, a.radices);
    if (a.n_radices < 0 || avg < 1 || frames_n < 0 || n_out < 1 || new_0 < 0 || spare_n < 0 || spare_begin < 0 ||
        !in || !window || !tw || !out || (new_0 > 0 && !carry) || (spare_n > 0 && !carry_out) || (!sel && n_out != nfft))
        return hipErrorInvalidValue;
    // every read stays inside the logical window [carry | in]
    if ((frames_n > 0 && (long long)(frames_n + avg - 1) * nfft > window_len) ||
        (long long)spare_begin + spare_n > window_len || new_0 > window_len)
        return hipErrorInvalidValue;
    a.carry = carry; a.in = in; a.window = window; a.tw = tw; a.sel = sel; a.out = out; a.carry_out = carry_out;
    a.n = nfft; a.F = avg; a.frames_n = frames_n; a.n_out = n_out; a.new_0 = new_0;
    a.spare_begin = spare_begin; a.spare_n = spare_n;
    // short frames share a workgroup: at least ~1024 points of work per workgroup
    a.FR = nfft >= 1024 ? 1 : (1024 + nfft - 1) / nfft;
    if (a.FR > 64) a.FR = 64;
    a.main_blocks = (unsigned)((frames_n + a.FR - 1) / a.FR);
    if (window_len > 0x7fffffffLL - nfft) return hipErrorInvalidValue;     // 32-bit window positions in the kernel
    auto magic = [](long long d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / (unsigned long long)d + 1ULL); };
    a.mag_n = magic(nfft);
    a.mag_nout = magic(n_out);
    {
        int p = 1;
        for (int s = 0; s < a.n_radices; ++s) {
            a.mag_t[s] = magic(nfft / a.radices[s]);
            a.mag_p[s] = magic(p);
            a.stage_t[s] = nfft / a.radices[s];
            a.stage_tws[s] = nfft / (p * a.radices[s]);
            p *= a.radices[s];
        }
        if (a.n_radices > 0 && a.radices[0] > 13) {
            const int t4 = (nfft / a.radices[0] + 3) / 4;
            a.mag_t4 = magic(t4);
        }
    }
    a.blocks_per_xcd = (a.main_blocks + 7u) / 8u;
    const unsigned spare_blocks = (unsigned)((spare_n + 2047) / 2048);
    if (a.main_blocks + spare_blocks == 0) return hipSuccess;
    const unsigned grid = a.blocks_per_xcd * 8u + spare_blocks;
    const bool twl = nfft <= kPfbLdsTwMaxN;
    const size_t lds = ((size_t)2 * a.FR * nfft + kPfbLdsMaxPrime + 1 + (twl ? nfft : 0)) * sizeof(float2);   // two frame sets + roots (+ twiddles)
    // (a function attribute belongs to the device it was set on: once per device of this process)
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
    const bool wide = nfft >= 2048;                      // 512 threads per frame
    const void *fn = twl ? (wide ? reinterpret_cast<const void *>(pfb_lds_kernel<true, 512>)
                                 : reinterpret_cast<const void *>(pfb_lds_kernel<true, 256>))
                         : (wide ? reinterpret_cast<const void *>(pfb_lds_kernel<false, 512>)
                                 : reinterpret_cast<const void *>(pfb_lds_kernel<false, 256>));
    if (!(attr_done.load() >> dev & 1ULL)) {
        for (const void *f : {reinterpret_cast<const void *>(pfb_lds_kernel<true, 256>),
                              reinterpret_cast<const void *>(pfb_lds_kernel<true, 512>),
                              reinterpret_cast<const void *>(pfb_lds_kernel<false, 256>),
                              reinterpret_cast<const void *>(pfb_lds_kernel<false, 512>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kPfbLdsMaxBytes);
            if (e != hipSuccess) return e;
        }
        attr_done.fetch_or(1ULL << dev);
    }
    if (lds > (size_t)kPfbLdsMaxBytes) return hipErrorInvalidValue;
    void *kargs[] = {&a};
    return hipLaunchKernel(fn, dim3(grid), dim3(wide ? 512 : 256), kargs, lds, st);
}

const char *pfb_lds_kernel_name() { return "pfb_lds_kernel"; }
const char *fft_kernel_name() { return "fft_pass_kernel"; }

}  // namespace gsdr
