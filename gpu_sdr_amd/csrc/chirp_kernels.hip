// chirp_kernels.hip -- chirp (VNA) demodulator, fused lock-in decimator and
// the synthetic IQ sources, for gfx950.
//
// ref: chirp_demodulator cpp/kernels.cu:389-427, cublas_decim (Cgemv lock-in)
// cpp/kernels.cu:852-872, chirp_gen cpp/kernels.cu:335-372.
//
// The reference demodulates the whole buffer to memory and then runs a GEMV
// over it (two passes + a move_buffer carry).  Here the lock-in path is ONE
// pass: each wave owns one output point, demodulates its ppt samples in
// registers and reduces them with cross-lane shuffles; only raw samples are
// carried between buffers (demodulation is a pure function of the chirp index).
// HBM-bound: 8 B read per sample, 8/ppt B written.
#include <hip/hip_runtime.h>

#include "ddc_kernels.h"

namespace gsdr {

// sin/cos of pi*index/2147483647.5 (ref: kernels.cu:421-422) from the int32
// index.  pi*index/2147483647.5 == 2*pi*index/(2^32-1); replacing 2^32-1 by
// 2^32 moves the angle by < 2^-33 turn (7e-10 rad), far below float
// resolution, and makes the range reduction exact integer arithmetic.
__device__ __forceinline__ void sincos_index(int index, float &s, float &c) {
    const unsigned u = (unsigned)index;
    const unsigned quad = (u + 0x20000000u) >> 30;          // nearest quarter turn
    const int r = (int)(u - (quad << 30));                   // [-2^29, 2^29)
    const float a = (float)r * 1.4629180792671596e-09f;     // 2*pi / 2^32, |a| <= pi/4
    const float a2 = a * a;
    float sp = fmaf(a2, 2.7557319223985893e-06f, -1.9841269841269841e-04f);
    sp = fmaf(a2, sp, 8.3333333333333333e-03f);
    sp = fmaf(a2, sp, -1.6666666666666666e-01f);
    const float sn = fmaf(a * a2, sp, a);
    float cp = fmaf(a2, -2.7557319223985888e-07f, 2.4801587301587302e-05f);
    cp = fmaf(a2, cp, -1.3888888888888889e-03f);
    cp = fmaf(a2, cp, 4.1666666666666664e-02f);
    cp = fmaf(a2, cp, -0.5f);
    const float cn = fmaf(a2, cp, 1.0f);
    switch (quad & 3u) {
        case 0: s = sn;  c = cn;  break;
        case 1: s = cn;  c = -sn; break;
        case 2: s = -sn; c = -cn; break;
        default: s = -cn; c = sn; break;
    }
}

// ref: kernels.cu:407-419.  The reference evaluates this in 64-bit unsigned
// arithmetic and truncates to int; only the low 32 bits survive, so when the
// step index fits 32 bits everything can be done modulo 2^32.
__device__ __forceinline__ int chirp_index(unsigned long long e, const ChirpShape &cs,
                                           bool small) {
    if (small) {
        const unsigned e32 = (unsigned)e;
        const unsigned fi = e32 / (unsigned)cs.length;
        const unsigned q = (fi >> 1) * (fi + 1u) + (fi & 1u) * ((fi + 1u) >> 1);
        const unsigned idx = e32 * ((unsigned)cs.f0 + fi * cs.chirpness) -
                             cs.chirpness * ((unsigned)cs.length * q);
        return (int)idx;
    }
    const unsigned long long fi = e / cs.length;
    const unsigned long long q = (fi / 2) * (fi + 1) + (fi % 2) * ((fi + 1) / 2);
    const unsigned long long corr = (unsigned long long)cs.chirpness * (cs.length * q);
    const unsigned long long idx =
        e * ((unsigned long long)(long long)cs.f0 + fi * cs.chirpness) - corr;
    return (int)idx;
}

__device__ __forceinline__ float2 demod_one(float2 in, int index) {
    float s, c;
    sincos_index(index, s, c);
    const float chx = s, chy = -c;  // chirp = (sinpi, -cospi), kernels.cu:421-422
    float2 o;                        // in * conj(chirp), kernels.cu:424-425
    o.x = chx * in.x + chy * in.y;
    o.y = chx * in.y - chy * in.x;
    return o;
}

__global__ __launch_bounds__(256) void chirp_demod_kernel(const float2 *__restrict__ in,
                                                          float2 *__restrict__ out, long long n,
                                                          unsigned long long index0,
                                                          ChirpShape cs) {
    const bool small = cs.period < 0xffffffffull && cs.num_steps < 0xfffffffeull;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += stride) {
        unsigned long long e;
        if (small) {
            unsigned r = (unsigned)((unsigned long long)o % (unsigned)cs.period);
            unsigned long long t = index0 + r;
            e = t >= cs.period ? t - cs.period : t;
        } else {
            e = (index0 + (unsigned long long)o) % cs.period;
        }
        out[o] = demod_one(in[o], chirp_index(e, cs, small));
    }
}

// One wave per output point v (4 per workgroup).
__global__ __launch_bounds__(256) void chirp_lockin_kernel(
    const float2 *__restrict__ carry, int carry_len, const float2 *__restrict__ in,
    const float *__restrict__ profile, int ppt, int valid, float2 *__restrict__ out,
    unsigned long long index0, ChirpShape cs) {
    const bool small = cs.period < 0xffffffffull && cs.num_steps < 0xfffffffeull;
    const int lane = threadIdx.x & 63;
    const int v = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (v >= valid) return;
    const long long base = (long long)v * ppt;  // position in the logical stage
    const unsigned long long e0 = (index0 + (unsigned long long)base) % cs.period;
    float sx = 0.f, sy = 0.f;
    for (int p = lane; p < ppt; p += 64) {
        const float w = profile[p];
        const long long pos = base + p;
        const float2 s = pos < carry_len ? carry[pos] : in[pos - carry_len];
        unsigned long long e = e0 + (unsigned long long)p;
        if (e >= cs.period) e = (e - cs.period < cs.period) ? e - cs.period : e % cs.period;
        const float2 d = demod_one(s, chirp_index(e, cs, small));
        sx = fmaf(d.x, w, sx);
        sy = fmaf(d.y, w, sy);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_xor(sx, off, 64);
        sy += __shfl_xor(sy, off, 64);
    }
    if (lane == 0) out[v] = make_float2(sx, sy);
}

hipError_t launch_chirp_demod(const float2 *in, float2 *out, long long n,
                              unsigned long long index0, const ChirpShape &cs, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(chirp_demod_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n,
                       index0, cs);
    return hipGetLastError();
}

hipError_t launch_chirp_lockin(const float2 *carry, int carry_len, const float2 *in,
                               const float *profile, int ppt, int valid, float2 *out,
                               unsigned long long index0, const ChirpShape &cs, hipStream_t st) {
    if (valid <= 0) return hipSuccess;
    hipLaunchKernelGGL(chirp_lockin_kernel, dim3((unsigned)((valid + 3) / 4)), dim3(256), 0, st,
                       carry, carry_len, in, profile, ppt, valid, out, index0, cs);
    return hipGetLastError();
}

const char *chirp_demod_kernel_name() { return "chirp_demod_kernel"; }
const char *chirp_lockin_kernel_name() { return "chirp_lockin_kernel"; }

// ---------------------------------------------------------------------------
// synthetic sources (benchmark input; replace the UHD/sw-loop RX threads,
// ref: cpp/USRP_hardware_manager.cpp:1331-1395)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void source_tones_kernel(
    float2 *__restrict__ out, long long n, long long start, unsigned rate,
    const unsigned *__restrict__ fmod, const float *__restrict__ ampl,
    const float *__restrict__ phase, int n_tones, float sigma, unsigned long long seed) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const double inv_rate = 1.0 / (double)rate;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const unsigned long long s = (unsigned long long)(start + j) % rate;
        float re = 0.f, im = 0.f;
        for (int k = 0; k < n_tones; ++k) {
            // exact integer phase (f*s mod rate), then one float sincos
            const unsigned long long ph = ((unsigned long long)fmod[k] * s) % rate;
            const float turns2 = (float)(2.0 * ((double)ph * inv_rate));
            float sn, cs;
            sincospif(turns2, &sn, &cs);
            float s0, c0;
            sincosf(phase[k], &s0, &c0);
            const float a = ampl[k];
            re += a * (cs * c0 - sn * s0);
            im += a * (sn * c0 + cs * s0);
        }
        if (sigma > 0.f) {
            const unsigned long long h = mix64(seed ^ mix64((unsigned long long)(start + j)));
            const float u1 = ((float)(unsigned)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);
            const float u2 = (float)(unsigned)((h >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
            const float rad = sigma * sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincospif(2.0f * u2, &sn, &cs);
            re += rad * cs;
            im += rad * sn;
        }
        out[j] = make_float2(re, im);
    }
}

__global__ __launch_bounds__(256) void source_chirp_kernel(float2 *__restrict__ out, long long n,
                                                           unsigned long long index0,
                                                           ChirpShape cs, float scale) {
    const bool small = cs.period < 0xffffffffull && cs.num_steps < 0xfffffffeull;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += stride) {
        const unsigned long long e = (index0 + (unsigned long long)o) % cs.period;
        float s, c;
        sincos_index(chirp_index(e, cs, small), s, c);
        out[o] = make_float2(s * scale, -c * scale);  // ref: kernels.cu:367-368
    }
}

hipError_t launch_source_tones(float2 *out, long long n, long long start, unsigned rate,
                               const unsigned *fmod_dev, const float *ampl_dev,
                               const float *phase_dev, int n_tones, float sigma,
                               unsigned long long seed, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(source_tones_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, n,
                       start, rate, fmod_dev, ampl_dev, phase_dev, n_tones, sigma, seed);
    return hipGetLastError();
}

hipError_t launch_source_chirp(float2 *out, long long n, unsigned long long index0,
                               const ChirpShape &cs, float scale, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(source_chirp_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, n,
                       index0, cs, scale);
    return hipGetLastError();
}

}  // namespace gsdr
