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
#include <cstdlib>
#include <vector>

#include "ddc_kernels.h"

namespace gsdr {

// The GSDR_PFB_* switches below exist for A/B runs and for the tests that drive every kernel variant.  They are read
// once and cached (a launch must not walk the environment); gsdr_reload_env() (include/gsdr.h) makes the next use
// read them again.
static std::atomic<int> g_env_generation{0};
void fft_env_reload() { g_env_generation.fetch_add(1); }
namespace {
struct EnvSwitch {
    const char *name;
    int unset;                 // value when the variable is not set (or empty)
    std::atomic<int> seen{-1}, value{0};
    EnvSwitch(const char *n, int u) : name(n), unset(u) {}
    int get() {
        const int g = g_env_generation.load(std::memory_order_relaxed);
        if (seen.load(std::memory_order_acquire) != g) {
            const char *e = std::getenv(name);
            value.store(e && e[0] ? std::atoi(e) : unset, std::memory_order_relaxed);
            seen.store(g, std::memory_order_release);
        }
        return value.load(std::memory_order_relaxed);
    }
};
EnvSwitch env_radix8{"GSDR_PFB_RADIX8", 1}, env_direct{"GSDR_PFB_DIRECT", 1}, env_col{"GSDR_PFB_COL", -1},
    env_cu_nt{"GSDR_PFB_CU_NT", 0}, env_cu{"GSDR_PFB_CU", -1}, env_fr{"GSDR_PFB_FR", 0}, env_wide{"GSDR_PFB_WIDE", -1},
    env_teams{"GSDR_PFB_TEAMS", 1};
}  // namespace

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
    } else if constexpr (R == 8) {
        // DFT-8 = two DFT-4 (even / odd inputs) + four constant twiddles: X[k] = E[k] + w8^k O[k], X[k+4] = E[k] - w8^k O[k]
        float2 e[4] = {u[0], u[2], u[4], u[6]}, o[4] = {u[1], u[3], u[5], u[7]};
        butterfly<4>(e, tw, n);
        butterfly<4>(o, tw, n);
        constexpr float h = 0.70710678118654752440f;
        const float2 t0 = o[0];
        const float2 t1 = mk2((o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h);      // * (1 - i) / sqrt 2
        const float2 t2 = mk2(o[2].y, -o[2].x);                                   // * -i
        const float2 t3 = mk2((o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h);     // * (-1 - i) / sqrt 2
        const float2 t[4] = {t0, t1, t2, t3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u[k] = mk2(e[k].x + t[k].x, e[k].y + t[k].y);
            u[k + 4] = mk2(e[k].x - t[k].x, e[k].y - t[k].y);
        }
    } else if constexpr (R == 6 || R == 10) {
        // DFT-2P (P = 3, 5) = two DFT-P (even / odd inputs) + constant twiddles: X[k] = E[k mod P] + w_2P^k O[k mod P]
        constexpr int P = R / 2;
        float2 e[P], o[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            e[i] = u[2 * i];
            o[i] = u[2 * i + 1];
        }
        butterfly<P>(e, tw, n);
        butterfly<P>(o, tw, n);
        // w_2P^k = (cos, -sin)(2 pi k / 2P), k < P (the other half is its negative)
        constexpr float c6[3] = {1.f, 0.5f, -0.5f}, s6[3] = {0.f, 0.86602540378443864676f, 0.86602540378443864676f};
        constexpr float c10[5] = {1.f, 0.80901699437494742410f, 0.30901699437494742410f, -0.30901699437494742410f, -0.80901699437494742410f};
        constexpr float s10[5] = {0.f, 0.58778525229247312917f, 0.95105651629515357212f, 0.95105651629515357212f, 0.58778525229247312917f};
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float wr = P == 3 ? c6[k] : c10[k], wi = -(P == 3 ? s6[k] : s10[k]);
            const float2 t = mk2(o[k].x * wr - o[k].y * wi, o[k].x * wi + o[k].y * wr);
            // X[k] = E[k] + t, X[k + P] = E[k] - t   (w_2P^(k+P) = -w_2P^k; (k + P) mod P = k)
            u[k] = mk2(e[k].x + t.x, e[k].y + t.y);
            u[k + P] = mk2(e[k].x - t.x, e[k].y - t.y);
        }
    } else if constexpr (R == 16) {
        // DFT-16 as 4 x 4 (round 3): r = r0 + 4 r1, q = q1 + 4 q0,
        //     X[q1 + 4 q0] = sum_r0 w4^(q0 r0) [ w16^(q1 r0) sum_r1 w4^(q1 r1) x[r0 + 4 r1] ]
        // -- two levels of radix-4 butterflies in registers with nine constant twiddles between them: one LDS (or
        // memory) round trip, one barrier and one set of index arithmetic where two radix-4 stages have two
        float2 y[4][4];
#pragma unroll
        for (int r0 = 0; r0 < 4; ++r0) {
            float2 v[4] = {u[r0], u[r0 + 4], u[r0 + 8], u[r0 + 12]};
            butterfly<4>(v, tw, n);
#pragma unroll
            for (int q1 = 0; q1 < 4; ++q1) y[r0][q1] = v[q1];
        }
        // w16^m = (cos, -sin)(2 pi m / 16), m = q1 r0
        constexpr float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, h = 0.70710678118654752440f;
        auto mulc = [](float2 a, float wr, float wi) { return mk2(a.x * wr - a.y * wi, a.x * wi + a.y * wr); };
        y[1][1] = mulc(y[1][1], c1, -s1);      // m = 1
        y[1][2] = mulc(y[1][2], h, -h);        // m = 2
        y[1][3] = mulc(y[1][3], s1, -c1);      // m = 3
        y[2][1] = mulc(y[2][1], h, -h);        // m = 2
        y[2][2] = mk2(y[2][2].y, -y[2][2].x);  // m = 4: -i
        y[2][3] = mulc(y[2][3], -h, -h);       // m = 6
        y[3][1] = mulc(y[3][1], s1, -c1);      // m = 3
        y[3][2] = mulc(y[3][2], -h, -h);       // m = 6
        y[3][3] = mulc(y[3][3], -c1, s1);      // m = 9
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) {
            float2 v[4] = {y[0][q1], y[1][q1], y[2][q1], y[3][q1]};
            butterfly<4>(v, tw, n);
#pragma unroll
            for (int q0 = 0; q0 < 4; ++q0) u[q1 + 4 * q0] = v[q0];
        }
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
__device__ unsigned g_fft_stamp_mask = 0xffu;       // which slots are written (the stamps cost time themselves: A/B with 0x81)
__device__ __forceinline__ void fft_stamp(int slot) {
    if (g_fft_stamp_buf && threadIdx.x == 0 && (g_fft_stamp_mask >> slot & 1u))
        g_fft_stamp_buf[8 * (size_t)blockIdx.x + slot] = __builtin_amdgcn_s_memrealtime();
}
// core clocks between two points (s_memtime counts shader clocks): (slot, c0, 0) remembers, (slot, c0, 1) stores the difference
__device__ __forceinline__ void fft_stamp_core(int slot, unsigned long long &c0, int end) {
    if (!end) c0 = __builtin_amdgcn_s_memtime();
    else if (g_fft_stamp_buf && threadIdx.x == 0) g_fft_stamp_buf[8 * (size_t)blockIdx.x + slot] = __builtin_amdgcn_s_memtime() - c0;
}
#else
__device__ __forceinline__ void fft_stamp(int) {}
__device__ __forceinline__ void fft_stamp_core(int, unsigned long long &, int) {}
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
    unsigned long long core0 = 0;
    fft_stamp_core(6, core0, 0);
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
            case 16: lds_stage<16>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 8: lds_stage<8>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 6: lds_stage<6>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 10: lds_stage<10>(src, dst, n, p, st, stw, mt, mp, tw, FR, tid, NT); break;
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
    fft_stamp_core(6, core0, 1);
    fft_stamp(7);
}


// ---------------------------------------------------------------------------------------------
// Round 3: a RUN of consecutive frames per workgroup, one workgroup (1024 threads) per compute unit.
//
// pfb_lds_kernel above gives every frame a workgroup of its own; all workgroups of a launch start together
// and stay in step: six microseconds in which every compute unit waits for its filter loads, then five in
// which the memory system idles while they all transform (profiles/r02_stamp_pfb.log: 785 GB/s = 0.098 of the
// HBM peak).  Frame r and frame r+1 share F-1 of their F blocks, and each of those workgroups pulled all F
// through its own 8-byte loads: 32 load instructions per thread and round trip.
//
// Here the grid is one workgroup per compute unit (G = frames / 256 consecutive frames each: ~3 900 samples of
// new data per unit at any frame length), and a workgroup
//   1. stages the raw samples of its run ONCE -- (G + F - 1) n samples, 16 bytes per lane and load, four or five
//      loads per thread, one memory round trip -- into the LDS (the region the transform's second buffer uses
//      later), the window taps of its points into registers on the same trip;
//   2. filters out of the LDS (float accumulation in tap order, as the reference's kernel,
//      ref: cpp/kernels.cu:474-516);
//   3. transforms all G frames together: a stage is one butterfly per thread over the whole run, so the
//      barrier-separated stages are as many as for one frame;
//   4. selects the bins and stores.
// A prime factor above 13 is the first stage as before, one output pair per work item now (item = frame, q,
// column: four times the items of the four-column version, a quarter of the chain each).
// Frame lengths with a prime factor above 127 (or any length, GSDR_PFB_BLUESTEIN=1) go through Bluestein's
// identity INSIDE the workgroup: chirp, zero padding to m = 2^ceil(log2(2n-1)) <= 8192, two radix-4/2
// transforms of length m around a pointwise product with the chirp's transform (host-made, double), chirp
// again -- the reference takes any fft_tones through one cufftPlanMany (ref: cpp/USRP_demodulator.cpp:150-153).
// ---------------------------------------------------------------------------------------------
constexpr int kPfbCuThreads = 1024;
constexpr int kPfbCuPts = 6;                     // filter points per thread at most: G * n <= 6 * 1024
constexpr int kPfbCuMaxBytes = 156 * 1024;       // of the 160 KiB of a compute unit

struct PfbCuArgs {
    const float2 *carry, *in;
    const float *window;
    const float2 *tw;          // w_len^k, k < len
    const int *sel;
    float2 *out, *carry_out;
    const float2 *chirp;       // Bluestein: exp(+i pi k^2 / n), k < n; nullptr: the frame length is transformed directly
    const float2 *bhat;        // Bluestein: the transform (length m) of the wrapped chirp
    int n, F, frames_n, n_out, new_0, G, len;    // len: transform length (n, or m)
    int spare_begin, spare_n;
    unsigned main_blocks, blocks_per_xcd;
    int n_radices;
    int radices[16];
    unsigned mag_n, mag_nout, mag_len, mag_pad, mag_ht;
    unsigned mag_t[16], mag_p[16];
    int stage_t[16], stage_tws[16];
    int b_off, b_len;          // second buffer: offset and length (float2), also holds the raw samples first
    int twl;                   // the twiddle table goes into the LDS too
    int col;                   // the filter works column-wise (see the kernel)
    int direct;                // ... and straight out of global memory: 1: <1, 11>, 2: <2, 7>, 3: <4, 4> (columns per thread, blocks), 4: <1, 11> in groups
    int dir_s, dir_gs;         // frames shorter than the workgroup: dir_s groups of threads take dir_gs consecutive frames each
    int teams;                 // the stages of a frame run on a team of NT / G threads of its own (see pfb_team_barrier)
};

// the prime-first stage with one (q, column) pair per work item; see lds_stage_prime_first
__device__ __forceinline__ void lds_stage_prime_first1(int R, float2 *src, float2 *dst, int n, const float2 *roots,
                                                       int t, unsigned mag_t, unsigned mag_ht, int FR, int tid, int NT) {
    const int h = (R - 1) >> 1;
    for (int g = tid; g < FR * h * t; g += NT) {
        const int rr = fdiv(g, mag_t), i = g - rr * t;          // rr = fr * h + (r - 1)
        const int fr = FR == 1 ? 0 : rr / h, r = rr - fr * h + 1;
        const int lo = fr * n + i + r * t, hi = fr * n + i + (R - r) * t;
        const float2 x = src[lo], y = src[hi];
        src[lo] = mk2(x.x + y.x, x.y + y.y);
        src[hi] = mk2(x.x - y.x, x.y - y.y);
    }
    __syncthreads();
    const int per = (h + 1) * t;
    for (int g = tid; g < FR * per; g += NT) {
        const int fr = FR == 1 ? 0 : fdiv(g, mag_ht), rem = g - fr * per;
        const int q = fdiv(rem, mag_t), i = rem - q * t;
        const float2 *xb = src + fr * n + i;
        const float2 x0 = xb[0];
        float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
        int idx = 0;
#pragma unroll 4
        for (int r = 1; r <= h; ++r) {
            idx += q;
            idx = idx >= R ? idx - R : idx;
            const float2 w = roots[idx];             // (cos, -sin)
            const float2 S = xb[r * t], D = xb[(R - r) * t];
            ax = fmaf(S.x, w.x, ax);
            ay = fmaf(S.y, w.x, ay);
            bx = fmaf(D.x, -w.y, bx);
            by = fmaf(D.y, -w.y, by);
        }
        const int o = fr * n + i * R;
        const float cx = x0.x + ax, cy = x0.y + ay;
        dst[o + q] = mk2(cx + by, cy - bx);
        if (q != 0) dst[o + R - q] = mk2(cx - by, cy + bx);
    }
}

// The prime-first stage as two small dense real matrix products on the matrix cores (v_mfma_f32_16x16x4_f32: exact
// f32 products, f32 accumulate -- this IS a dense contraction, unlike the rest of the path).  After the S / D pass
// (see lds_stage_prime_first) the stage is
//     A[q][c] = sum_(r=1..h) cos(2 pi q r / R) S_r[c],    B[q][c] = sum_(r=1..h) sin(2 pi q r / R) D_r[c],
// q = 0 .. h, over the columns c = (frame, column i, re | im) -- 2 FR t real columns -- and
//     out[q] = x_0 + A - iB,   out[R - q] = x_0 + A + iB.
// On the VALU that is an instruction per multiply-add plus the LDS reads and index arithmetic around it:
// 7.2 us of a 1230-point run of four frames (R = 41), 14.6 us of a 1016-point run (R = 127) -- more than everything
// else of those launches together (profiles/r03_stamp_pfb_cu.log).  A wave takes 16 x 16 tiles of (q, c): per step
// of four r one root (cos, -sin) and one S and one D value per lane out of the LDS, two MFMAs.
typedef float pfb_f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_stage_prime_first_mfma(int R, const float2 *src, float2 *dst, int n, const float2 *roots,
                                                           int t, unsigned mag_t, int FR, int tid, int NT) {
    // S_r = x_r + x_(R-r) and D_r = x_r - x_(R-r) are formed where they are used (two LDS reads and two additions
    // per operand: a pass of its own that forms them in place cost a microsecond and a barrier).  Tiles of 16 x 16
    // (v_mfma_f32_16x16x4_f32, four r per step): R = 127 over a run of four 1016-point frames is 4 x 4 tiles --
    // every wave of the workgroup has one -- where 32 x 32 tiles left all but two or four waves idle.
    const int h = (R - 1) >> 1;
    const float *sf = reinterpret_cast<const float *>(src);
    float *df = reinterpret_cast<float *>(dst);
    const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
    const int l16 = lane & 15, l4 = lane >> 4;
    const int NC = 2 * FR * t;                                  // real columns: (frame, i, re | im)
    const int MT = (h + 1 + 15) >> 4, NTL = (NC + 15) >> 4;
    const int steps = (h + 3) >> 2;                             // four r per step; r > h meets a zero coefficient
    // tile -> (mt, nt) without a division: MT <= 4 (R <= kPfbLdsMaxPrime = 127)
    for (int tile = wave; tile < MT * NTL; tile += nwaves) {
        int mt = 0, nt = tile;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (nt >= NTL) {
                nt -= NTL;
                ++mt;
            }
        // this lane's row of the coefficient matrices and its column of S / D
        const int q = mt * 16 + l16, qc = q <= h ? q : h;
        int c = nt * 16 + l16;
        const bool col_ok = c < NC;
        c = col_ok ? c : NC - 1;
        const int part = c & 1, ci = c >> 1;
        const int fr = FR == 1 ? 0 : fdiv(ci, mag_t), i = ci - fr * t;
        const int col0 = 2 * (fr * n + i) + part;               // float index of x_0's component
        int idx = qc * (1 + l4);                                // <= 4 h < 2 R
        idx = idx >= R ? idx - R : idx;
        int dq = 4 * qc;
        dq = dq >= R ? dq - R : dq;
        int r = 1 + l4;
        pfb_f4v accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
        // Four steps -- eight MFMAs -- make a batch, and the operands of a batch live in registers of their own
        // (two sets, used in turn): an MFMA reads its A and B registers pass by pass while it runs (rule R1 of
        // DESIGN.md section 4.1).  The matrix pipe takes its instructions in order: once the eight MFMAs of batch b
        // have been ISSUED, those of batch b-1 are complete, so the loads of batch b+1 may then land in b-1's set --
        // and their LDS latency hides under the 256 cycles batch b is running.  `pin` keeps the compiler from using a
        // set's registers for anything else before that point.
        struct Ops {
            float ac[4], as[4], bs[4], bd[4];
        };
        auto load = [&](Ops &o) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + 4 * u;
                const bool live = ru <= h;
                const int rc = live ? ru : h;                   // steps past the last: a valid address, a zero coefficient
                const float2 w = roots[idx];                    // (cos, -sin)(2 pi q r / R)
                const float xl = sf[col0 + 2 * rc * t], xh = sf[col0 + 2 * (R - rc) * t];
                o.ac[u] = live ? w.x : 0.f;
                o.as[u] = live ? -w.y : 0.f;
                o.bs[u] = xl + xh;                              // S_r
                o.bd[u] = xl - xh;                              // D_r
                idx += dq;
                idx = idx >= R ? idx - R : idx;
            }
            r += 16;
        };
        auto mma = [&](const Ops &o) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ac[u], o.bs[u], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(o.as[u], o.bd[u], accB, 0, 0, 0);
            }
        };
        auto pin = [&](const Ops &o) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                asm volatile("" ::"v"(o.ac[u]), "v"(o.as[u]), "v"(o.bs[u]), "v"(o.bd[u]));
        };
        Ops oa, ob;
        load(oa);
        for (int s = 0; s < steps; s += 8) {
            mma(oa);
            load(ob);
            pin(oa);
            mma(ob);
            load(oa);
            pin(ob);
        }
        pin(oa);
        if (tile == 0) fft_stamp(6);
        // The results first, then their readers: behind the compiler's own wait count VALU reads of an MFMA's
        // result registers may see stale values (the hazard ddc_mfma_kernel guards against, csrc/ddc_mfma.hip).
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        // register j of a lane: row 4 (lane / 16) + j of the tile, column lane % 16;
        //     out[q] = x_0 + A - iB,   out[R - q] = x_0 + A + iB
        // The re and im columns of a point sit in neighbouring lanes: the cross terms are one DPP swap away.
        const float x0p = sf[col0];
        const float sgn = part ? -1.f : 1.f;
        const int obase = 2 * (fr * n + i * R) + part;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = mt * 16 + 4 * l4 + j;
            // (through a scalar copy: __builtin_bit_cast applied to the vector ELEMENT accB[j] reads element 0
            //  whatever j is)
            const float bown = accB[j];
            const float bp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bown), 0xB1, 0xf, 0xf, false));
            const float base = x0p + accA[j];
            if (col_ok && qq <= h) {
                df[obase + 2 * qq] = base + sgn * bp;
                if (qq != 0) df[obase + 2 * (R - qq)] = base - sgn * bp;
            }
        }
    }
}

// The same stage on 32 x 32 tiles (v_mfma_f32_32x32x2_f32) with a POINT (frame, i) per column and its real and
// imaginary parts as two accumulations side by side: no lane exchange, an output is one 8-byte store.  Fewer, fatter
// tiles: the better choice when the run has many columns (R = 41 over four 1230-point frames: 120 points, stage
// 4.3 us against 7.7 us on 16 x 16 tiles, whose per-tile set-up is paid 30 times there); with few columns (R = 127:
// 32 points) it leaves all but two waves idle and the 16 x 16 form wins (7.2 against 8.7 us).
typedef float pfb_f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void lds_stage_prime_first_mfma32(int R, const float2 *src, float2 *dst, int n, const float2 *roots,
                                                             int t, unsigned mag_t, int FR, int tid, int NT) {
    const int h = (R - 1) >> 1;
    const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
    const int l32 = lane & 31, l2 = lane >> 5;
    const int NC = FR * t;                                      // columns: the points (frame, i)
    const int MT = (h + 1 + 31) >> 5, NTL = (NC + 31) >> 5;     // MT <= 2: R <= kPfbLdsMaxPrime = 127
    const int steps = (h + 1) >> 1;                             // two r per step; r > h meets a zero coefficient
    for (int tile = wave; tile < MT * NTL; tile += nwaves) {
        const int mt = tile >= NTL ? 1 : 0, nt = tile - mt * NTL;
        // this lane's row of the coefficient matrices (q * r < R for q <= h, r <= 2: no reduction needed at the start)
        const int q = mt * 32 + l32, qc = q <= h ? q : h;
        int c = nt * 32 + l32;
        const bool col_ok = c < NC;
        c = col_ok ? c : NC - 1;
        const int fr = FR == 1 ? 0 : fdiv(c, mag_t), i = c - fr * t;
        const float2 *xc = src + fr * n + i;                    // x_r of this column: xc[r * t]
        int idx = qc * (1 + l2);
        const int dq = 2 * qc;                                  // < R
        int r = 1 + l2;
        pfb_f16v aRe = {0}, aIm = {0}, bRe = {0}, bIm = {0};
        // batches of two steps (eight MFMAs), two operand sets: see lds_stage_prime_first_mfma
        struct Ops {
            float ac[2], as[2], sre[2], sim[2], dre[2], dim[2];
        };
        auto load = [&](Ops &o) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ru = r + 2 * u;
                const bool live = ru <= h;
                const int rc = live ? ru : h;                   // steps past the last: a valid address, a zero coefficient
                const float2 w = roots[idx];                    // (cos, -sin)(2 pi q r / R)
                const float2 xl = xc[rc * t], xh = xc[(R - rc) * t];
                o.ac[u] = live ? w.x : 0.f;
                o.as[u] = live ? -w.y : 0.f;
                o.sre[u] = xl.x + xh.x;                         // S_r
                o.sim[u] = xl.y + xh.y;
                o.dre[u] = xl.x - xh.x;                         // D_r
                o.dim[u] = xl.y - xh.y;
                idx += dq;
                idx = idx >= R ? idx - R : idx;
            }
            r += 4;
        };
        auto mma = [&](const Ops &o) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(o.ac[u], o.sre[u], aRe, 0, 0, 0);
                aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(o.ac[u], o.sim[u], aIm, 0, 0, 0);
                bRe = __builtin_amdgcn_mfma_f32_32x32x2f32(o.as[u], o.dre[u], bRe, 0, 0, 0);
                bIm = __builtin_amdgcn_mfma_f32_32x32x2f32(o.as[u], o.dim[u], bIm, 0, 0, 0);
            }
        };
        auto pin = [&](const Ops &o) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                asm volatile("" ::"v"(o.ac[u]), "v"(o.as[u]), "v"(o.sre[u]), "v"(o.sim[u]), "v"(o.dre[u]), "v"(o.dim[u]));
        };
        Ops oa, ob;
        load(oa);
        for (int s = 0; s < steps; s += 4) {
            mma(oa);
            load(ob);
            pin(oa);
            mma(ob);
            load(oa);
            pin(ob);
        }
        pin(oa);
        if (tile == 0) fft_stamp(6);
        // the results first, then their readers (eighty cycles: the last MFMA runs 64)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        // register j of a lane: row 8 (j / 4) + 4 l2 + j % 4 of the tile, column l32:
        //     out[q] = x_0 + A - iB,   out[R - q] = x_0 + A + iB
        const float2 x0 = xc[0];
        const int qb = mt * 32 + 4 * l2;                        // the lane's rows: qb + 8 (j / 4) + j % 4
        float2 *oq = dst + fr * n + i * R + qb;                 // out[qb + off]
        float2 *om = dst + fr * n + i * R + (R - qb);           // out[R - qb - off]
        const int room = col_ok ? h - qb : -1;                  // rows with off <= room exist
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int off = 8 * (j >> 2) + (j & 3);
            const float cx = x0.x + aRe[j], cy = x0.y + aIm[j];
            const float bx = bRe[j], by = bIm[j];
            if (off <= room) {
                oq[off] = mk2(cx + by, cy - bx);
                if (qb + off != 0) om[-off] = mk2(cx - by, cy + bx);
            }
        }
    }
}

// all stages of a.radices over FR frames of `len` points; returns where the result is
// (buffers are named by their OFFSET in the LDS array, not by pointer: a pointer that is swapped in a loop or
//  picked by a comparison loses its address space and every ds_read behind it turns into a flat_load)
template <int NT>
__device__ __forceinline__ int pfb_cu_stages(const PfbCuArgs &a, float2 *lds, int src_off, int dst_off, const float2 *tw,
                                             const float2 *roots, int st_radix, int st_mag_t, int st_mag_p,
                                             int st_t, int st_tws, int tid, int s0, int s1, int &p) {
    const int len = a.len, FR = a.G;
    for (int s = s0; s < s1; ++s) {
        float2 *src = lds + src_off, *dst = lds + dst_off;
        const int R = __builtin_amdgcn_readlane(st_radix, s);
        const unsigned mt = (unsigned)__builtin_amdgcn_readlane(st_mag_t, s), mp = (unsigned)__builtin_amdgcn_readlane(st_mag_p, s);
        const int st = __builtin_amdgcn_readlane(st_t, s), stw = __builtin_amdgcn_readlane(st_tws, s);
        switch (R) {
            case 2: lds_stage<2>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 3: lds_stage<3>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 4: lds_stage<4>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 5: lds_stage<5>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 7: lds_stage<7>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 11: lds_stage<11>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 13: lds_stage<13>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 16: lds_stage<16>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 8: lds_stage<8>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 6: lds_stage<6>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            case 10: lds_stage<10>(src, dst, len, p, st, stw, mt, mp, tw, FR, tid, NT); break;
            default:
                if (s == 0 && FR * st >= 64)            // many columns: fat tiles, a point per column
                    lds_stage_prime_first_mfma32(R, src, dst, len, roots, st, mt, FR, tid, NT);
                else if (s == 0)
                    lds_stage_prime_first_mfma(R, src, dst, len, roots, st, mt, FR, tid, NT);
                else
                    lds_stage_generic(R, src, dst, len, p, tw, FR, tid, NT);
                break;
        }
        __syncthreads();
        if (s < 2) fft_stamp(3 + s);
        p *= R;
        const int t2 = src_off;
        src_off = dst_off;
        dst_off = t2;
    }
    return src_off;
}

// The polyphase filter of a run straight out of global memory, for four taps: a thread takes CPT columns k = tid + NT c
// and loads, per column, the NB >= Gw + 3 blocks the run's frames are made of -- ONE load per (block, column) where
// the frame-per-workgroup kernel issues four (every frame reloads its three shared blocks) and the staged path of this
// kernel puts a trip through the LDS and a barrier in between -- then forms every frame of the run from registers
// (float accumulation in tap order, ref cpp/kernels.cu:474-516).  All loads of a thread are in flight together.
template <int CPT, int NB, bool GROUPS, int NT>
__device__ __forceinline__ void pfb_cu_filter_direct(const PfbCuArgs &a, float2 *A, int f0, int Gw, int tid) {
    const int n = a.n, len = a.len;
    // GROUPS (frames of at most half the workgroup): the threads form dir_s groups of n, group g takes the dir_gs
    // consecutive frames from g * dir_gs on (their dir_gs + 3 blocks); otherwise one group takes the whole run -- a
    // template parameter, so that without groups every condition on a frame or block number stays wave-uniform
    const int grp = GROUPS ? fdiv(tid, a.mag_n) : 0;
    const int k0 = tid - grp * n;
    const int fa = grp * a.dir_gs;                         // first frame of this thread's group, within the run
    const int gf = a.G - fa < a.dir_gs ? a.G - fa : a.dir_gs;       // frames of the group (<= 0: none)
    const int gw = Gw - fa;                                // ... of which exist (may be <= 0)
    float2 xr[CPT][NB];
    float w[CPT][4];
    float2 ch[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int k = k0 + NT * c;
        const bool live = k < n && grp < a.dir_s && gw > 0;  // (columns beyond the frame, groups without a frame: no loads)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[c][j] = live && j < a.F ? a.window[(unsigned)(j * n + k)] : 0.f;
        ch[c] = mk2(1.f, 0.f);
        if (a.chirp && live) {
            const float2 cc = a.chirp[k];
            ch[c] = mk2(cc.x, -cc.y);
        }
        const int q0 = (f0 + fa) * n + k;                  // window position of block 0 of this column
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            xr[c][b] = mk2(0.f, 0.f);
            if (live && b < gw + a.F - 1) {      // (up to four taps: with fewer the last taps are zero and their blocks stay unloaded, zero)
                const int q = q0 + b * n;
                xr[c][b] = q < a.new_0 ? a.carry[q] : a.in[q - a.new_0];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int k = k0 + NT * c;
        if (k < n && grp < a.dir_s) {
#pragma unroll
            for (int fr = 0; fr < NB - 3; ++fr) {
                if (fr < gf) {
                    float2 acc = mk2(0.f, 0.f);
                    if (fr < gw) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i < a.F) {                 // (a later frame's block never enters this one: 0 * Inf)
                                acc.x += xr[c][fr + i].x * w[c][i];
                                acc.y += xr[c][fr + i].y * w[c][i];
                            }
                        }
                        if (a.chirp) acc = cmul(acc, ch[c]);
                    }
                    A[(fa + fr) * len + k] = acc;
                }
            }
        }
    }
}

// TWL: the twiddle table is copied into the LDS (a template parameter, not a run-time choice between an LDS and a
// global pointer: that would be a flat pointer)
// A barrier among the waves of ONE team (the NT / G threads that take one frame of the run through its stages).
// The workgroup's frames are independent from the filter on; s_barrier makes all sixteen waves meet after every
// stage, in step -- every wave then fights for the LDS and the issue slots at the same time and idles together
// (1.75 us per radix-8 stage against 1.15 us in four separate workgroups of 256 threads).  gfx950 has
// no named barriers, so a team counts arrivals in an LDS word: the wave's own LDS traffic has completed
// (release fence = s_waitcnt lgkmcnt(0)), one lane adds 1, everyone polls until all `waves` arrivals of this round
// are in.  The count only grows (round r ends at r * waves), nothing is reset, nobody can be left behind.  The poll
// is bounded: a team that never completes (it cannot, short of a bug) falls through instead of hanging the chip.
__device__ __forceinline__ void pfb_team_barrier(unsigned *ctr, unsigned &target, unsigned waves, int tid) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    target += waves;
    if ((tid & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int guard = 0; guard < (1 << 20); ++guard) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// one stage of one frame on a team (the radices of plans without a matrix-core stage)
__device__ __forceinline__ void pfb_team_stage(int R, const float2 *src, float2 *dst, int n, int p, int st, int stw, unsigned mt,
                                               unsigned mp, const float2 *tw, int ttid, int TS) {
    switch (R) {
        case 2: lds_stage<2>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 3: lds_stage<3>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 4: lds_stage<4>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 5: lds_stage<5>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 6: lds_stage<6>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 7: lds_stage<7>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 8: lds_stage<8>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 10: lds_stage<10>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 11: lds_stage<11>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        case 13: lds_stage<13>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
        default: lds_stage<16>(src, dst, n, p, st, stw, mt, mp, tw, 1, ttid, TS); break;
    }
}

// NT: 1024 threads, one workgroup per compute unit -- or 512, two per unit with half the frames each (the direct
// filter only: it needs no raw samples in the LDS): two workgroups drift apart, one computes while the other waits
// at a barrier or for its loads, where the sixteen waves of one workgroup meet at every barrier in step
template <bool TWL, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) GSDR_NO_PK void pfb_cu_kernel(const PfbCuArgs a) {
    extern __shared__ float2 pfb_lds[];
    const int tid = threadIdx.x, n = a.n, len = a.len, G = a.G;
    const unsigned padded = a.blocks_per_xcd * 8u;
    if (blockIdx.x >= padded) {
        // leftovers of this call -> carry of the next one (PfbLdsArgs has the same fields: one helper)
        const int j0 = (int)(blockIdx.x - padded) * 2048;
        for (int j = j0 + tid; j < j0 + 2048 && j < a.spare_n; j += NT) {
            const int q = a.spare_begin + j;
            a.carry_out[j] = q < a.new_0 ? a.carry[q] : a.in[q - a.new_0];
        }
        return;
    }
    // contiguous runs of frames per XCD (its L2 then serves the blocks neighbouring runs share)
    const unsigned wg = (blockIdx.x & 7u) * a.blocks_per_xcd + (blockIdx.x >> 3);
    if (wg >= a.main_blocks) return;
    const int f0 = (int)wg * G;
    const int Gw = a.frames_n - f0 < G ? a.frames_n - f0 : G;       // frames of this run that exist (>= 1)
    fft_stamp(0);
    float2 *A = pfb_lds, *B = pfb_lds + a.b_off, *roots = B + a.b_len, *twl = roots + (kPfbLdsMaxPrime + 1);
    float2 *raw = B;
    // the parameters of stage s live in lane s of five registers
    const int st_lane = tid & 15;
    const int st_radix = a.radices[st_lane], st_mag_t = (int)a.mag_t[st_lane], st_mag_p = (int)a.mag_p[st_lane];
    const int st_t = a.stage_t[st_lane], st_tws = a.stage_tws[st_lane];
    const int R0 = a.n_radices > 0 && a.radices[0] > 13 ? a.radices[0] : 0;

    int *sel_l = reinterpret_cast<int *>(twl + (TWL ? len : 0));
    if (a.direct) {
        // ---- 1 + 2 in one: tables into the LDS, the filter straight out of global memory (pfb_cu_filter_direct) ----
        if (TWL)
            for (int k = tid; k < len; k += NT) twl[k] = a.tw[k];
        for (int m = tid; m < R0; m += NT) roots[m] = a.tw[m * (len / R0)];
        if (a.sel)
            for (int u = tid; u < a.n_out; u += NT) sel_l[u] = a.sel[u];
        switch (a.direct) {
            case 1: pfb_cu_filter_direct<1, 11, false, NT>(a, A, f0, Gw, tid); break;
            case 2: pfb_cu_filter_direct<2, 7, false, NT>(a, A, f0, Gw, tid); break;
            case 3: pfb_cu_filter_direct<4, 4, false, NT>(a, A, f0, Gw, tid); break;
            case 5: pfb_cu_filter_direct<3, 5, false, NT>(a, A, f0, Gw, tid); break;
            default: pfb_cu_filter_direct<1, 11, true, NT>(a, A, f0, Gw, tid); break;
        }
        fft_stamp(1);
    } else {
    // ---- 1. loads: the window taps of this thread's points, then the raw samples of the run ----
    // F == 4 (the client's default): a thread filters COLUMNS k = tid + NT c of all G frames -- consecutive frames
    // share F - 1 of their F blocks, so a column costs G + 3 LDS reads and four taps instead of 4 G and 4 G -- and
    // the taps below are those of its columns.  Other F: a thread filters points p = tid + NT c = (frame, k).
    const bool col_mode = a.col != 0;
    int pk[kPfbCuPts], pfr[kPfbCuPts];
    float wv[kPfbCuPts][4];
    const int npts = G * n;
#pragma unroll
    for (int c = 0; c < kPfbCuPts; ++c) {
        const int p = tid + NT * c, pc = p < npts ? p : 0;
        pfr[c] = col_mode ? 0 : fdiv(pc, a.mag_n);
        pk[c] = col_mode ? (p < n ? p : 0) : pc - pfr[c] * n;
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[c][j] = a.window[(unsigned)((j < a.F ? j : a.F - 1) * n + pk[c])];
    }
    const int nraw = (Gw + a.F - 1) * n;
    const int q0 = f0 * n;                                  // window position of raw[0]
    if (q0 >= a.new_0) {
        const float2 *src = a.in + (q0 - a.new_0);
        typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
        typedef float f4a __attribute__((ext_vector_type(4)));
        for (int j = 2 * tid; j < nraw; j += 2 * NT) {
            if (j + 1 < nraw) {
                const f4u v = *reinterpret_cast<const f4u *>(src + j);
                *reinterpret_cast<f4a *>(raw + j) = f4a{v.x, v.y, v.z, v.w};
            } else {
                raw[j] = src[j];
            }
        }
    } else {
        for (int j = tid; j < nraw; j += NT) {
            const int q = q0 + j;
            raw[j] = q < a.new_0 ? a.carry[q] : a.in[q - a.new_0];
        }
    }
    if (TWL)
        for (int k = tid; k < len; k += NT) twl[k] = a.tw[k];
    for (int m = tid; m < R0; m += NT) roots[m] = a.tw[m * (len / R0)];
    // the bin of every output column, read behind the last stage: out of the LDS (64 clocks), not out of the L2
    if (a.sel)
        for (int u = tid; u < a.n_out; u += NT) sel_l[u] = a.sel[u];
    __syncthreads();
    fft_stamp(1);

    // ---- 2. polyphase filter out of the LDS (float accumulate in tap order) ----
    if (col_mode) {
        // the frame loop is the outer one so that every register array keeps compile-time indices
        float2 x0[kPfbCuPts], x1[kPfbCuPts], x2[kPfbCuPts], chv[kPfbCuPts];
#pragma unroll
        for (int c = 0; c < kPfbCuPts; ++c) {
            const int k = tid + NT * c;
            x0[c] = x1[c] = x2[c] = mk2(0.f, 0.f);
            chv[c] = mk2(1.f, 0.f);
            if (k < n) {
                x0[c] = raw[k];
                x1[c] = raw[n + k];
                x2[c] = raw[2 * n + k];
                if (a.chirp) {
                    const float2 cc = a.chirp[k];
                    chv[c] = mk2(cc.x, -cc.y);
                }
            }
        }
        for (int fr = 0; fr < G; ++fr) {
            const bool live = fr < Gw;
            const float2 *rp = raw + (fr + 3) * n;
            float2 *ap = A + fr * len;
#pragma unroll
            for (int c = 0; c < kPfbCuPts; ++c) {
                const int k = tid + NT * c;
                if (k < n) {
                    float2 acc = mk2(0.f, 0.f);
                    if (live) {
                        const float2 x3 = rp[k];
                        acc.x += x0[c].x * wv[c][0];
                        acc.y += x0[c].y * wv[c][0];
                        acc.x += x1[c].x * wv[c][1];
                        acc.y += x1[c].y * wv[c][1];
                        acc.x += x2[c].x * wv[c][2];
                        acc.y += x2[c].y * wv[c][2];
                        acc.x += x3.x * wv[c][3];
                        acc.y += x3.y * wv[c][3];
                        x0[c] = x1[c];
                        x1[c] = x2[c];
                        x2[c] = x3;
                        if (a.chirp) acc = cmul(acc, chv[c]);
                    }
                    ap[k] = acc;
                }
            }
        }
    } else {
#pragma unroll
    for (int c = 0; c < kPfbCuPts; ++c) {
        const int p = tid + NT * c;
        if (p < npts) {
            float2 acc = mk2(0.f, 0.f);
            if (pfr[c] < Gw) {
                const float2 *rp = raw + pfr[c] * n + pk[c];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < a.F) {
                        const float2 sm = rp[i * n];
                        acc.x += sm.x * wv[c][i];
                        acc.y += sm.y * wv[c][i];
                    }
                for (int i = 4; i < a.F; ++i) {
                    const float2 sm = rp[i * n];
                    const float w = a.window[(unsigned)(i * n + pk[c])];
                    acc.x += sm.x * w;
                    acc.y += sm.y * w;
                }
                if (a.chirp) {
                    const float2 ch = a.chirp[pk[c]];
                    acc = cmul(acc, mk2(ch.x, -ch.y));
                }
            }
            A[pfr[c] * len + pk[c]] = acc;
        }
    }
    }
    }   // (!a.direct)
    if (len > n) {                                          // Bluestein: zero padding up to m
        const int pad = len - n;
        for (int g = tid; g < G * pad; g += NT) {
            const int fr = fdiv(g, a.mag_pad);
            A[fr * len + n + (g - fr * pad)] = mk2(0.f, 0.f);
        }
    }
    // arrival counters of the teams (behind the bin table: (n + 1) / 2 float2 hold it)
    unsigned *team_ctr = reinterpret_cast<unsigned *>(sel_l + 2 * ((n + 1) / 2));
    if (a.teams && tid < G) team_ctr[tid] = 0u;
    __syncthreads();
    fft_stamp(2);

    const float2 *tw = TWL ? twl : a.tw;
    if (a.teams) {
        // ---- 3 + 4 by teams: NT / G threads take one frame through all stages and write its bins ----
        const int TS = NT / G, fr = tid / TS, ttid = tid - fr * TS;       // (G divides NT: the host only sets `teams` then)
        const unsigned waves = (unsigned)(TS >> 6);
        unsigned target = 0;
        int src_off = fr * len, dst_off = a.b_off + fr * len, p = 1;
        for (int s = 0; s < a.n_radices; ++s) {
            const int R = __builtin_amdgcn_readlane(st_radix, s);
            const unsigned mt = (unsigned)__builtin_amdgcn_readlane(st_mag_t, s), mp = (unsigned)__builtin_amdgcn_readlane(st_mag_p, s);
            const int st = __builtin_amdgcn_readlane(st_t, s), stw = __builtin_amdgcn_readlane(st_tws, s);
            pfb_team_stage(R, pfb_lds + src_off, pfb_lds + dst_off, len, p, st, stw, mt, mp, tw, ttid, TS);
            if (waves > 1) pfb_team_barrier(team_ctr + fr, target, waves, tid);
            else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (s < 2) fft_stamp(3 + s);
            p *= R;
            const int t2 = src_off;
            src_off = dst_off;
            dst_off = t2;
        }
        fft_stamp(5);
        if (fr < Gw) {
            const float2 *res = pfb_lds + src_off;
            float2 *o = a.out + (size_t)(f0 + fr) * a.n_out;
            for (int u = ttid; u < a.n_out; u += TS) o[u] = res[a.sel ? sel_l[u] : u];
        }
        fft_stamp(7);
        return;
    }

    // ---- 3. the transform of all G frames, stage by stage ----
    int pp = 1;
    int res_off = pfb_cu_stages<NT>(a, pfb_lds, 0, a.b_off, tw, roots, st_radix, st_mag_t, st_mag_p, st_t, st_tws, tid, 0, 1, pp);
    res_off = pfb_cu_stages<NT>(a, pfb_lds, res_off, res_off == 0 ? a.b_off : 0, tw, roots, st_radix, st_mag_t, st_mag_p, st_t,
                            st_tws, tid, 1, a.n_radices, pp);
    if (a.chirp) {
        // d = conj(A * Bhat); the inverse transform is then a forward one (IFFT(z) = conj(FFT(conj z)) / m)
        float2 *res = pfb_lds + res_off;
        for (int g = tid; g < G * len; g += NT) {
            const int fr = fdiv(g, a.mag_len);
            const float2 v = cmul(res[g], a.bhat[g - fr * len]);
            res[g] = mk2(v.x, -v.y);
        }
        __syncthreads();
        pp = 1;
        res_off = pfb_cu_stages<NT>(a, pfb_lds, res_off, res_off == 0 ? a.b_off : 0, tw, roots, st_radix, st_mag_t, st_mag_p,
                                st_t, st_tws, tid, 0, a.n_radices, pp);
    }
    const float2 *res = pfb_lds + res_off;
    fft_stamp(5);

    // ---- 4. bin selection and output ----
    const float inv_m = 1.f / (float)len;
#define GSDR_PFB_CU_EMIT(g, bin_expr)                                                         \
    {                                                                                         \
        const int fr = fdiv((g), a.mag_nout), u = (g)-fr * a.n_out;                           \
        if (fr < Gw) {                                                                        \
            const int bin = (bin_expr);                                                       \
            float2 v = res[fr * len + bin];                                                   \
            if (a.chirp) { /* X[k] = conj(chirp[k]) * conj(e[k]) / m */                       \
                const float2 ch = a.chirp[bin];                                               \
                const float2 r = cmul(mk2(ch.x, -ch.y), mk2(v.x, -v.y));                      \
                v = mk2(r.x * inv_m, r.y * inv_m);                                            \
            }                                                                                 \
            a.out[(size_t)(f0 + fr) * a.n_out + u] = v;                                       \
        }                                                                                     \
    }
    for (int g = tid; g < G * a.n_out; g += NT) GSDR_PFB_CU_EMIT(g, a.sel ? sel_l[u] : u)
#undef GSDR_PFB_CU_EMIT
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
        case 16: return launch_pass<16>(x, y, n, p, tw, batch, st);
        case 8: return launch_pass<8>(x, y, n, p, tw, batch, st);
        case 6: return launch_pass<6>(x, y, n, p, tw, batch, st);
        case 10: return launch_pass<10>(x, y, n, p, tw, batch, st);
        default: return hipErrorInvalidValue;
    }
}

// radices of n (16s, 4s, then 2, then odd primes up to 13); empty when a larger prime remains
std::vector<int> factorize(int n) {
    std::vector<int> r;
    int m = n;
    while (m % 16 == 0) { r.push_back(16); m /= 16; }      // (round 3) two radix-4 levels per pass through memory
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
extern "C" int gsdr_debug_set_fft_stamp_mask(unsigned mask) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fft_stamp_mask), &mask, sizeof(mask)) == hipSuccess ? 0 : -1;
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

// Stages of the in-LDS transform: 16s, 4s, 2, then every odd prime factor (3, 5, 7 with register
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
    // (round 3) Fewer, fatter stages -- every stage is a barrier, an LDS round trip and ~60 instructions of index
    // arithmetic per butterfly whatever its radix:
    //   * 16 (two radix-4 levels in registers) for frames of 4096 points and more: 4096 points 14.8 -> 13.3 us per
    //     buffer, 8192 27.0 -> 23.2.  Below that it does not pay: one thread in sixteen points busy leaves too few
    //     waves to hide each other's latencies (1024 points: two radix-16 stages take what four radix-4 stages took);
    //   * 8 (= 4 x 2) otherwise: one thread in eight points keeps half of the threads busy;
    //   * a single 2 left over joins a 3 or a 5: radix 6 / 10 (1230 = 41 * 6 * 5, 1000 = 8 * 5 * 5 * 5).
    // GSDR_PFB_RADIX8=0: 4s and 2s as in round 2 (A/B runs)
    const bool fat = env_radix8.get() != 0;
    if (n >= 4096)
        while (m % 16 == 0) { push(16); m /= 16; }
    if (fat)
        while (m % 8 == 0) { push(8); m /= 8; }
    while (m % 4 == 0) { push(4); m /= 4; }
    if (fat && m % 2 == 0 && m % 3 == 0) { push(6); m /= 6; }
    if (fat && m % 2 == 0 && m % 5 == 0) { push(10); m /= 10; }
    for (int q : {2, 3, 5, 7, 11, 13})
        while (m % q == 0) { push(q); m /= q; }
    return cnt <= 16 ? cnt : -1;
}

// Up to four taps and frames of 128 points and more: the run kernel filters straight out of global memory (one load per
// block and column instead of four, no trip through the LDS) and is the faster one for powers of two as well --
// same box, per 1 M-sample buffer: 256 points 10.1 against 10.3 us, 512: 10.2 / 10.8, 1024: 10.7 / 12.2, 2048:
// 10.4 / 11.9; 16 points 9.9 against 9.4 and 64 points equal: short frames stay a frame set per workgroup
// (profiles/r03_pfb_ab_direct.log).  GSDR_PFB_DIRECT=0 switches the direct filter off, and this rule with it.
static bool pfb_cu_direct_pays(int nfft, int avg) {
    const int direct_env = env_direct.get();
    return direct_env && avg >= 1 && avg <= 4 && nfft >= 128;
}

// Shape of the run-per-compute-unit kernel for frames of nfft points transformed at length `len` (nfft, or
// Bluestein's m): frames per workgroup G (at most `want`), the offset and size of the second LDS buffer, whether
// the twiddle table fits beside them.  False when not even one frame fits.
static bool pfb_cu_shape(int nfft, int avg, int len, int want, int &G, int &b_off, int &b_len, int &twl, size_t &bytes,
                         int threads = kPfbCuThreads, long long max_bytes = kPfbCuMaxBytes, bool staged = true) {
    for (G = want < 1 ? 1 : want; G >= 1; --G) {
        if (staged && (long long)G * nfft > (long long)kPfbCuPts * threads) continue;
        const long long al = ((long long)G * len + 1) & ~1LL;                 // even: 16-byte LDS stores into the buffer behind
        long long bl = staged ? (long long)(G + avg - 1) * nfft : 0;          // (the direct filter stages no raw samples)
        if (bl < (long long)G * len) bl = (long long)G * len;
        bl = (bl + 1) & ~1LL;
        for (twl = len <= kPfbLdsTwMaxN ? 1 : 0; twl >= 0; --twl) {
            // + the bin table (n_out <= nfft ints)
            const long long total = (al + bl + kPfbLdsMaxPrime + 1 + (twl ? len : 0) + (nfft + 1) / 2 + 32) * (long long)sizeof(float2);   // (+ 64 team counters)
            if (total <= max_bytes) {
                b_off = (int)al;
                b_len = (int)bl;
                bytes = (size_t)total;
                return true;
            }
        }
    }
    return false;
}

// Long frames, few of them: the run kernel gives every compute unit ceil(frames / units) frames, and with 325 frames
// of 3072 points that is two for 163 units and none for the rest (21 us per buffer against 15 a frame per workgroup,
// two workgroups per unit; profiles/r03_pfb_ab_4096.log).  The share of (unit, frame) slots a launch fills; below
// 0.7 the frame-per-workgroup kernel takes the call where it can (no matrix-core stage, no Bluestein).
static double pfb_cu_fill(int nfft, int avg, int len, int frames_n, int cus) {
    if (frames_n <= 0 || cus <= 0) return 1.0;
    const int want = (frames_n + cus - 1) / cus;
    int G = 0, bo, bl, twl;
    size_t bytes;
    if (!pfb_cu_shape(nfft, avg, len, want, G, bo, bl, twl, bytes) &&
        !pfb_cu_shape(nfft, avg, len, want, G, bo, bl, twl, bytes, kPfbCuThreads, kPfbCuMaxBytes, false))
        return 1.0;
    const long long blocks = (frames_n + G - 1) / G, rounds = (blocks + cus - 1) / cus;
    return (double)frames_n / ((double)G * cus * rounds);
}
static bool pfb_cu_has_big_prime(int nfft) {
    int r[16];
    const int nr = pfb_lds_plan(nfft, r);
    return nr > 0 && r[0] > 13;
}

bool pfb_cu_fits(int nfft, int avg, int len) {
    int G, bo, bl, twl;
    size_t bytes;
    if (!(nfft >= 1 && avg >= 1 && len >= nfft && len <= kPfbLdsMaxN)) return false;
    if (pfb_cu_shape(nfft, avg, len, 1, G, bo, bl, twl, bytes)) return true;
    // the direct filter keeps no raw samples in the LDS: a frame of up to four columns per thread fits without them
    return pfb_cu_direct_pays(nfft, avg) && nfft <= 4 * kPfbCuThreads &&
           pfb_cu_shape(nfft, avg, len, 1, G, bo, bl, twl, bytes, kPfbCuThreads, kPfbCuMaxBytes, false);
}

static hipError_t launch_pfb_cu(const float2 *carry, int new_0, const float2 *in, const float *window, const float2 *tw,
                                int nfft, int avg, int frames_n, const int *sel, int n_out, float2 *out,
                                float2 *carry_out, int spare_begin, int spare_n, const FftPlan *blue, hipStream_t st,
                                bool &taken) {
    taken = false;
    PfbCuArgs a{};
    const int len = blue ? blue->m : nfft;
    if (blue) {
        if (blue->n != nfft || blue->m < 2 * nfft - 1 || blue->n_radices > 16 || !blue->d_chirp || !blue->d_bhat)
            return hipErrorInvalidValue;
        a.n_radices = blue->n_radices;
        for (int i = 0; i < a.n_radices; ++i) a.radices[i] = blue->radices[i];
    } else {
        a.n_radices = pfb_lds_plan(nfft, a.radices);
        if (a.n_radices < 0) return hipSuccess;            // not this kernel's length
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return hipErrorInvalidDevice;
    static std::atomic<int> cu_count[64];
    if (cu_count[dev].load() == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cu_count[dev].store(cus);
    }
    cus = cu_count[dev].load();
    // one workgroup per compute unit: G consecutive frames each (frames_n == 0: a launch that only copies the carry)
    int want = frames_n > 0 ? (frames_n + cus - 1) / cus : 1;
    size_t lds = 0;
    if (n_out > nfft) return hipSuccess;                  // the bin table in the LDS is sized for n_out <= nfft
    // the direct filter's variant for G frames on `threads` threads: 0 = none
    // (GSDR_PFB_DIRECT=0: staged through the LDS; GSDR_PFB_COL=0/1, GSDR_PFB_CU_NT=512/1024: A/B runs)
    const int direct_env = env_direct.get();
    const int col_env = env_col.get() < 0 ? -1 : (env_col.get() != 0);
    const int nt_env = env_cu_nt.get();
    auto direct_variant = [&](int G, int threads, int &dir_s, int &dir_gs) {
        const int cpt = (nfft + threads - 1) / threads;
        dir_s = cpt == 1 ? threads / nfft : 1;             // groups of threads (frames shorter than the workgroup)
        dir_gs = (G + dir_s - 1) / dir_s;                  // frames per group
        const int nb = dir_gs + 3;                         // (the register shapes are those of four taps)
        if (avg < 1 || avg > 4 || !direct_env) return 0;
        if (cpt == 1 && nb <= 11) return dir_s > 1 ? 4 : 1;
        if (cpt <= 2 && nb <= 7) return 2;
        if (cpt <= 3 && nb <= 5) return 5;
        if (cpt <= 4 && nb <= 4) return 3;
        return 0;
    };
    int threads = kPfbCuThreads;
    a.direct = 0;
    // two workgroups of 512 threads per unit, half the frames each, when the direct filter takes them -- for frames
    // below 1024 points without a matrix-core stage: same box, 128 ... 512 points 9.7 - 10.0 against 10.2 - 10.3 us,
    // 1000: 12.4 / 12.9; from 1024 points on the full workgroup wins (1024: 10.9 against 11.2, 2048: 10.5 / 11.9,
    // and the matrix-core stage wants its sixteen waves: 1230: 13.5 / 17.0), profiles/r03_pfb_ab_nt.log
    const bool half_pays = nt_env == 512 || (nt_env == 0 && nfft < 1024 && !(a.n_radices > 0 && a.radices[0] > 13));
    if (half_pays && avg >= 1 && avg <= 4 && direct_env) {
        const int want2 = frames_n > 0 ? (frames_n + 2 * cus - 1) / (2 * cus) : 1;
        int G2 = 0, bo = 0, bl = 0, twl2 = 0, ds = 1, dg = 1;
        size_t lds2 = 0;
        if (pfb_cu_shape(nfft, avg, len, want2, G2, bo, bl, twl2, lds2, 512, kPfbCuMaxBytes / 2, false)) {
            const int v = direct_variant(G2, 512, ds, dg);
            if (v) {
                threads = 512;
                a.G = G2; a.b_off = bo; a.b_len = bl; a.twl = twl2; lds = lds2;
                a.direct = v; a.dir_s = ds; a.dir_gs = dg; a.col = 1;
            }
        }
    }
    if (threads == kPfbCuThreads) {
        if (pfb_cu_shape(nfft, avg, len, want, a.G, a.b_off, a.b_len, a.twl, lds)) {
            // column-wise filter: four taps, and enough columns for every thread
            a.col = avg == 4 && (col_env < 0 ? nfft >= kPfbCuThreads / 2 : col_env == 1);
            a.direct = direct_variant(a.G, kPfbCuThreads, a.dir_s, a.dir_gs);
        } else {
            // no room for the raw samples of a run: the direct filter needs none (4096 points: a frame per unit)
            if (!pfb_cu_shape(nfft, avg, len, want, a.G, a.b_off, a.b_len, a.twl, lds, kPfbCuThreads, kPfbCuMaxBytes, false))
                return hipSuccess;                         // does not fit: the caller's other kernel
            a.col = 1;
            while (a.G >= 1 && !(a.direct = direct_variant(a.G, kPfbCuThreads, a.dir_s, a.dir_gs))) --a.G;
            if (a.G < 1) return hipSuccess;
        }
    }
    {
        const bool cu_forced = env_cu.get() == 1;
        if (!blue && !cu_forced && threads == kPfbCuThreads && !(a.n_radices > 0 && a.radices[0] > 13) &&
            pfb_cu_fill(nfft, avg, len, frames_n, cus) < 0.7)
            return hipSuccess;                             // the frame-per-workgroup kernel fills the chip better
    }
    {
        // teams: the frames of a run go through their stages independently (GSDR_PFB_TEAMS=0: all waves in step)
        bool plain = !blue && a.n_radices > 0;
        for (int i = 0; i < a.n_radices; ++i) plain = plain && a.radices[i] <= 16;
        a.teams = plain && env_teams.get() != 0 && a.G >= 2 && threads % a.G == 0 && (threads / a.G) % 64 == 0;
    }
    a.carry = carry; a.in = in; a.window = window; a.tw = tw; a.sel = sel; a.out = out; a.carry_out = carry_out;
    a.chirp = blue ? blue->d_chirp : nullptr;
    a.bhat = blue ? blue->d_bhat : nullptr;
    a.n = nfft; a.F = avg; a.frames_n = frames_n; a.n_out = n_out; a.new_0 = new_0; a.len = len;
    a.spare_begin = spare_begin; a.spare_n = spare_n;
    a.main_blocks = (unsigned)((frames_n + a.G - 1) / a.G);
    auto magic = [](long long d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / (unsigned long long)d + 1ULL); };
    a.mag_n = magic(nfft);
    a.mag_nout = magic(n_out);
    a.mag_len = magic(len);
    a.mag_pad = magic(len - nfft);
    {
        int p = 1;
        for (int s = 0; s < a.n_radices; ++s) {
            a.mag_t[s] = magic(len / a.radices[s]);
            a.mag_p[s] = magic(p);
            a.stage_t[s] = len / a.radices[s];
            a.stage_tws[s] = len / (p * a.radices[s]);
            p *= a.radices[s];
        }
        if (a.n_radices > 0 && a.radices[0] > 13) {
            const int R = a.radices[0];
            a.mag_ht = magic((long long)((R - 1) / 2 + 1) * (len / R));
        }
    }
    a.blocks_per_xcd = (a.main_blocks + 7u) / 8u;
    const unsigned spare_blocks = (unsigned)((spare_n + 2047) / 2048);
    taken = true;
    if (a.main_blocks + spare_blocks == 0) return hipSuccess;
    static std::atomic<unsigned long long> attr_done{0};
    if (!(attr_done.load() >> dev & 1ULL)) {
        for (const void *f : {reinterpret_cast<const void *>(pfb_cu_kernel<true, 1024>), reinterpret_cast<const void *>(pfb_cu_kernel<false, 1024>),
                              reinterpret_cast<const void *>(pfb_cu_kernel<true, 512>), reinterpret_cast<const void *>(pfb_cu_kernel<false, 512>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kPfbCuMaxBytes);
            if (e != hipSuccess) return e;
        }
        attr_done.fetch_or(1ULL << dev);
    }
    void *kargs[] = {&a};
    const void *fn = threads == 512 ? (a.twl ? reinterpret_cast<const void *>(pfb_cu_kernel<true, 512>) : reinterpret_cast<const void *>(pfb_cu_kernel<false, 512>))
                                    : (a.twl ? reinterpret_cast<const void *>(pfb_cu_kernel<true, 1024>) : reinterpret_cast<const void *>(pfb_cu_kernel<false, 1024>));
    return hipLaunchKernel(fn, dim3(a.blocks_per_xcd * 8u + spare_blocks), dim3(threads), kargs, lds, st);
}

hipError_t launch_pfb_lds(const float2 *carry, int new_0, const float2 *in, const float *window, const float2 *tw,
                          int nfft, int avg, int frames_n, const int *sel, int n_out, float2 *out,
                          float2 *carry_out, int spare_begin, int spare_n, long long window_len, hipStream_t st,
                          const FftPlan *blue) {
    if (avg < 1 || frames_n < 0 || n_out < 1 || new_0 < 0 || spare_n < 0 || spare_begin < 0 || nfft < 1 ||
        !in || !window || !tw || !out || (new_0 > 0 && !carry) || (spare_n > 0 && !carry_out) || (!sel && n_out != nfft))
        return hipErrorInvalidValue;
    // every read stays inside the logical window [carry | in]
    if ((frames_n > 0 && (long long)(frames_n + avg - 1) * nfft > window_len) ||
        (long long)spare_begin + spare_n > window_len || new_0 > window_len || window_len > 0x7fffffffLL - nfft)
        return hipErrorInvalidValue;
    // A run of frames per compute unit (round 3) when the run fits the LDS and the length has a stage other than
    // radix 2 / 4 / 8 / 16 -- a prime above 13 (its stage runs on the matrix cores there), 3, 5, 7 ... --, goes
    // through Bluestein, or has four taps and at least 128 points (the direct filter, pfb_cu_direct_pays()).
    // Measured per 1 M-sample buffer (profiles/r03_pfb_sweep.log): 1230 points 19.7 -> 13.1 us, 1016: 21.5 -> 13.5,
    // 1024: 12.6 -> 10.9.  GSDR_PFB_CU=0 / 1 forces.
    const int cu_mode = env_cu.get() < 0 ? -1 : (env_cu.get() != 0);
    bool cu_wanted = cu_mode == 1 || blue != nullptr;
    if (cu_mode < 0 && !blue) {
        int r[16];
        const int nr = pfb_lds_plan(nfft, r);
        for (int i = 0; i < nr; ++i) cu_wanted |= (r[i] != 4 && r[i] != 2 && r[i] != 16 && r[i] != 8);
        cu_wanted |= pfb_cu_direct_pays(nfft, avg);
    }
    if (cu_wanted) {
        bool taken = false;
        const hipError_t e = launch_pfb_cu(carry, new_0, in, window, tw, nfft, avg, frames_n, sel, n_out, out, carry_out,
                                           spare_begin, spare_n, blue, st, taken);
        if (e != hipSuccess || taken) return e;
        if (blue) return hipErrorInvalidValue;             // only the run kernel knows Bluestein's identity
    }
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
    const int fr_env = env_fr.get();      // A/B runs
    const int wide_env = env_wide.get();
    if (fr_env > 0 && (long long)fr_env * nfft <= 4096) a.FR = fr_env;
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
    const bool wide = wide_env >= 0 ? wide_env != 0 : nfft >= 2048;                      // 512 threads per frame
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
const char *pfb_cu_kernel_name() { return "pfb_cu_kernel"; }
// which of the two kernels launch_pfb_lds() runs for this shape (describe(), the profiler's name)
bool pfb_cu_takes(int nfft, int avg, int len, bool bluestein, int frames_per_call) {
    const int cu_mode = env_cu.get() < 0 ? -1 : (env_cu.get() != 0);
    if (!bluestein) {
        int r[16];
        const int nr = pfb_lds_plan(nfft, r);
        if (cu_mode == 0 || nr < 0) return false;
        bool wanted = cu_mode == 1 || pfb_cu_direct_pays(nfft, avg);
        for (int i = 0; i < nr; ++i) wanted |= (r[i] != 4 && r[i] != 2 && r[i] != 16 && r[i] != 8);
        if (!wanted) return false;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        if (cu_mode != 1 && nfft >= kPfbCuThreads && !pfb_cu_has_big_prime(nfft) && pfb_cu_fill(nfft, avg, len, frames_per_call, cus) < 0.7)
            return false;
    }
    return pfb_cu_fits(nfft, avg, len);
}

const char *fft_kernel_name() { return "fft_pass_kernel"; }

}  // namespace gsdr
