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
#include <cmath>
#include <cstdint>
#include <vector>

#include "ddc_kernels.h"

namespace gsdr {

// No packed FP32 in these kernels: a NOISE handle may run beside the matrix-core DDC of another handle
// (two front-ends on one GPU), and v_pk_*_f32 with a high-half broadcast is unreliable in a wave that
// shares its SIMD with an MFMA loop (rule R3, DESIGN.md section 4.1, tools/ubench_pk_hazard.hip).
#define GSDR_NO_PK __attribute__((target("no-packed-fp32-ops")))

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

hipError_t launch_pfb_filter(const float2 *raw, const float *window, int nfft, int avg, int frames_n, float2 *frames,
                             hipStream_t st) {
    if (nfft < 1 || avg < 1 || frames_n < 1 || !raw || !window || !frames) return hipErrorInvalidValue;
    const long long total = (long long)frames_n * nfft;
    hipLaunchKernelGGL(pfb_filter_kernel, dim3(grid_for(total)), dim3(256), 0, st, raw, window, nfft, avg, total, frames);
    return hipGetLastError();
}

const char *fft_kernel_name() { return "fft_pass_kernel"; }

}  // namespace gsdr
