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

#include <cstdlib>

#include "ddc_kernels.h"
#include "ddc_device.h"

namespace gsdr {

// (make_float2 of the HIP headers is not always_inline: inside a kernel with other target features --
//  GSDR_NO_PK -- it would stay a real call)
__device__ __forceinline__ float2 mk2c(float x, float y) {
    float2 v;
    v.x = x;
    v.y = y;
    return v;
}

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
    // quarter-turn rotation without branches (a switch here compiles to four divergent
    // branches per sample): quad 0: (sn, cn), 1: (cn, -sn), 2: (-sn, -cn), 3: (-cn, sn)
    const bool swap = (quad & 1u) != 0;
    const unsigned s_sign = (quad & 2u) << 30, c_sign = ((quad + 1u) & 2u) << 30;
    s = __uint_as_float(__float_as_uint(swap ? cn : sn) ^ s_sign);
    c = __uint_as_float(__float_as_uint(swap ? sn : cn) ^ c_sign);
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

// the 32-bit fast kernels cover num_steps, length and period below 2^31 / 2^32
__host__ __device__ inline bool chirp_fits_32(const ChirpShape &cs) {
    return cs.period < 0xffffffffull && cs.num_steps < 0x7fffffffull && cs.length < 0x800000ull;
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

// Sum over the 64 lanes of a wave with DPP row operations (a butterfly of __shfl_xor is six
// dependent ds_bpermute round trips through the LDS crossbar, ~0.7 us at the end of a wave that
// lives 3 us); the total arrives in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
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

__global__ __launch_bounds__(256) GSDR_NO_PK void chirp_demod_generic_kernel(const float2 *__restrict__ in,
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
__global__ __launch_bounds__(256) GSDR_NO_PK void chirp_lockin_generic_kernel(
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
    sx = wave_sum_to_lane63(sx);
    sy = wave_sum_to_lane63(sy);
    if (lane == 63) out[v] = mk2c(sx, sy);
}

// ---------------------------------------------------------------------------
// Fast paths (every realistic sweep: num_steps and the period fit 32 bits).
//
// Within one frequency step fi the chirp index is LINEAR in the sample:
//     index(fi*len + r) = K(fi) + r*A(fi)   (mod 2^32)
//     A = f0 + fi*c,   K = fi*len*A - c*len*q(fi),   q = fi(fi+1)/2
// and the lock-in windows are aligned to steps (ppt = len*decim and the stream
// position of every window is a multiple of ppt), so the per-sample 64-bit
// modulo/divide of the generic kernels reduces to one multiply-add; A and K are
// wave-uniform in the lock-in kernel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void step_coeffs(unsigned fi, const ChirpShape &cs, unsigned &A,
                                            unsigned &K) {
    const unsigned len = (unsigned)cs.length;
    const unsigned q = (fi >> 1) * (fi + 1u) + (fi & 1u) * ((fi + 1u) >> 1);
    A = (unsigned)cs.f0 + fi * cs.chirpness;
    K = fi * len * A - cs.chirpness * (len * q);
}

// (index0 + pos) mod period with 32-bit arithmetic (index0 < period < 2^32, pos < 2^32)
__device__ __forceinline__ unsigned wrap_index(unsigned long long index0, unsigned pos,
                                               const ChirpShape &cs) {
    const unsigned per = (unsigned)cs.period;
    const unsigned long long t = index0 + (unsigned long long)(pos % per);
    return (unsigned)(t >= per ? t - per : t);
}

// One wave per output point.  The samples are taken in stretches of 256 (4 per lane); the loads
// of a stretch are issued one stretch ahead -- those of the first one in front of the index
// arithmetic of the prologue (three integer divisions), which then runs in their shadow -- and
// carry no branch (clamped addresses, the weight of a lane beyond the step is zeroed by a select).
// With ppt = 200 (C4) a wave lives for one memory round trip: the launch is 5000 such waves.
struct ChirpStretch {
    float2 smp[4];
    float w[4];
};

__device__ __forceinline__ void load_stretch(ChirpStretch &c, const float2 *__restrict__ carry, int carry_len,
                                             const float2 *__restrict__ in, const float *__restrict__ w_seg,
                                             unsigned seg, unsigned r0, unsigned len, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned r = r0 + (unsigned)(i * 64 + lane);
        const bool ok = r < len;
        const unsigned rc = ok ? r : 0u;
        const unsigned pos = seg + rc;
        const float2 *p = (int)pos < carry_len ? carry + pos : in + (pos - (unsigned)carry_len);
        c.smp[i] = *p;
        c.w[i] = w_seg[rc];          // zeroed at the use (a select here would wait for the load)
    }
}

__global__ __launch_bounds__(256) GSDR_NO_PK void chirp_lockin_kernel(
    const float2 *__restrict__ carry, int carry_len, const float2 *__restrict__ in,
    const float *__restrict__ profile, int ppt, int decim, int valid, float2 *__restrict__ out,
    unsigned long long index0, ChirpShape cs) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int v = (int)blockIdx.x * 4 + wid;
    if (v >= valid) return;
    const unsigned len = (unsigned)cs.length, steps = (unsigned)cs.num_steps;
    const unsigned base = (unsigned)v * (unsigned)ppt;  // position in the logical stage [carry | in]
    ChirpStretch cur, nxt;
    load_stretch(cur, carry, carry_len, in, profile, base, 0u, len, lane);
    asm volatile("" ::: "memory");                      // the loads stay in front of the divisions
    const unsigned e0 = wrap_index(index0, base, cs);   // a multiple of len
    unsigned fi = e0 / len;
    float sx = 0.f, sy = 0.f;
    for (int d = 0; d < decim; ++d) {         // decim = ppt / len steps per point
        unsigned A, K;
        step_coeffs(fi, cs, A, K);
        float sd, cd;
        sincos_index((int)(64u * A), sd, cd);
        for (unsigned r0 = 0; r0 < len; r0 += 256) {
            // the stretch after this one: same step, or the head of the next step
            const bool more_here = r0 + 256 < len;
            if (more_here || d + 1 < decim) {
                const unsigned nd = more_here ? (unsigned)d : (unsigned)d + 1u;
                load_stretch(nxt, carry, carry_len, in, profile + (size_t)nd * len, base + nd * len,
                             more_here ? r0 + 256 : 0u, len, lane);
            }
            // the lane's four samples are 64 apart: inside a step the index advances by 64*A (mod 2^32),
            // the chirp phasor by one fixed rotation -- one sincos per lane and stretch instead of four
            float sn, cn;
            sincos_index((int)(K + (r0 + (unsigned)lane) * A), sn, cn);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned r = r0 + (unsigned)(i * 64 + lane);
                const float2 in = cur.smp[i];
                const float chx = sn, chy = -cn;     // chirp = (sinpi, -cospi); in * conj(chirp): demod_one()
                const float dx = chx * in.x + chy * in.y, dy = chx * in.y - chy * in.x;
                const float w = r < len ? cur.w[i] : 0.f;
                sx = fmaf(dx, w, sx);
                sy = fmaf(dy, w, sy);
                const float s2 = fmaf(sn, cd, cn * sd), c2 = fmaf(cn, cd, -(sn * sd));
                sn = s2;
                cn = c2;
            }
            cur = nxt;
        }
        fi = fi + 1 == steps ? 0 : fi + 1;
    }
    sx = wave_sum_to_lane63(sx);
    sy = wave_sum_to_lane63(sy);
    if (lane == 63) out[v] = mk2c(sx, sy);
}

// Many samples per point (round 3).  chirp_lockin_kernel gives a point to ONE wave: right for the benchmark's 200
// samples per point, but a VNA scan of a few hundred points over seconds has 1e4 ... 1e6 of them, and a 1 M-sample
// buffer then holds ten points, or one, or none -- ten waves, or one, walked the whole buffer (ppt 1e5: 162 us per
// buffer, ppt 1e6: 1.5 ms, profiles/r03_chirp_sweep.log).  Here the stretches of a point -- `decim` steps of
// ceil(len / 256) stretches each -- are dealt to `parts` waves in contiguous runs; every wave leaves the sum of its
// run in `partial[point * parts + part]`, chirp_lockin_sum_kernel adds a point's partial sums in order (no atomics:
// the result does not depend on which wave finishes first).  Same arithmetic per sample as chirp_lockin_kernel.
__global__ __launch_bounds__(256) GSDR_NO_PK void chirp_lockin_split_kernel(
    const float2 *__restrict__ carry, int carry_len, const float2 *__restrict__ in,
    const float *__restrict__ profile, int ppt, int decim, int valid, int parts, int per_part,
    float2 *__restrict__ partial, unsigned long long index0, ChirpShape cs) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave = (int)blockIdx.x * 4 + wid;
    if (wave >= valid * parts) return;
    const int v = wave / parts, part = wave - v * parts;     // wave-uniform
    const unsigned len = (unsigned)cs.length, steps = (unsigned)cs.num_steps;
    const unsigned nst = (len + 255u) / 256u, total = (unsigned)decim * nst;
    const unsigned q0 = (unsigned)part * (unsigned)per_part;
    const unsigned q1 = q0 + (unsigned)per_part < total ? q0 + (unsigned)per_part : total;
    float sx = 0.f, sy = 0.f;
    if (q0 < q1) {
        const unsigned base = (unsigned)v * (unsigned)ppt;  // position in the logical stage [carry | in]
        unsigned d = q0 / nst, r0 = (q0 - d * nst) * 256u;
        ChirpStretch cur, nxt;
        load_stretch(cur, carry, carry_len, in, profile + (size_t)d * len, base + d * len, r0, len, lane);
        asm volatile("" ::: "memory");                      // the loads stay in front of the divisions
        const unsigned e0 = wrap_index(index0, base, cs);   // a multiple of len
        unsigned fi = (e0 / len + d) % steps;
        bool new_step = true;
        unsigned A = 0, K = 0;
        float sd = 0.f, cd = 1.f;
        for (unsigned q = q0; q < q1; ++q) {
            if (new_step) {
                step_coeffs(fi, cs, A, K);
                sincos_index((int)(64u * A), sd, cd);
                new_step = false;
            }
            unsigned nd = d, nr0 = r0 + 256u;
            if (nr0 >= len) {
                nd = d + 1u;
                nr0 = 0u;
            }
            if (q + 1u < q1)
                load_stretch(nxt, carry, carry_len, in, profile + (size_t)nd * len, base + nd * len, nr0, len, lane);
            float sn, cn;
            sincos_index((int)(K + (r0 + (unsigned)lane) * A), sn, cn);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned r = r0 + (unsigned)(i * 64 + lane);
                const float2 smp = cur.smp[i];
                const float chx = sn, chy = -cn;     // chirp = (sinpi, -cospi); in * conj(chirp): demod_one()
                const float dx = chx * smp.x + chy * smp.y, dy = chx * smp.y - chy * smp.x;
                const float w = r < len ? cur.w[i] : 0.f;
                sx = fmaf(dx, w, sx);
                sy = fmaf(dy, w, sy);
                const float s2 = fmaf(sn, cd, cn * sd), c2 = fmaf(cn, cd, -(sn * sd));
                sn = s2;
                cn = c2;
            }
            cur = nxt;
            if (nd != d) {
                d = nd;
                fi = fi + 1u == steps ? 0u : fi + 1u;
                new_step = true;
            }
            r0 = nr0;
        }
    }
    sx = wave_sum_to_lane63(sx);
    sy = wave_sum_to_lane63(sy);
    if (lane == 63) partial[wave] = mk2c(sx, sy);
}

// a wave per point: lane l adds the partial sums l, l + 64, ... in order, the wave adds its lanes (a fixed tree)
__global__ __launch_bounds__(256) GSDR_NO_PK void chirp_lockin_sum_kernel(const float2 *__restrict__ partial, int valid, int parts,
                                                                          float2 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int v = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (v >= valid) return;
    const float2 *__restrict__ pp = partial + (size_t)v * parts;
    float sx = 0.f, sy = 0.f;
    for (int p = lane; p < parts; p += 64) {
        const float2 t = pp[p];
        sx += t.x;
        sy += t.y;
    }
    sx = wave_sum_to_lane63(sx);
    sy = wave_sum_to_lane63(sy);
    if (lane == 63) out[v] = mk2c(sx, sy);
}

// Undecimated demodulation: 4 consecutive runs of 64 samples per wave; the step
// index of a sample is the wave's base step plus a small quotient.
__global__ __launch_bounds__(256) GSDR_NO_PK void chirp_demod_kernel(const float2 *__restrict__ in,
                                                          float2 *__restrict__ out, long long n,
                                                          unsigned long long index0,
                                                          ChirpShape cs) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned len = (unsigned)cs.length, steps = (unsigned)cs.num_steps;
    const float inv_len = 1.0f / (float)len;
    const long long wave_stride = (long long)gridDim.x * 4 * 256;
    for (long long o0 = ((long long)blockIdx.x * 4 + wid) * 256; o0 < n; o0 += wave_stride) {
        const unsigned e_w = wrap_index(index0, (unsigned)o0, cs);
        const unsigned fi_w = e_w / len, r_w = e_w - fi_w * len;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long o = o0 + i * 64 + lane;
            if (o < n) {
                unsigned r = r_w + (unsigned)(i * 64 + lane);        // < len + 256
                unsigned qd = (unsigned)((float)r * inv_len);         // r / len, off by at most one
                unsigned rem = r - qd * len;
                if ((int)rem < 0) { qd--; rem += len; }
                if (rem >= len) { qd++; rem -= len; }
                unsigned fi = fi_w + qd;
                if (fi >= steps) fi %= steps;
                unsigned A, K;
                step_coeffs(fi, cs, A, K);
                out[o] = demod_one(in[o], (int)(K + rem * A));
            }
        }
    }
}

hipError_t launch_chirp_demod(const float2 *in, float2 *out, long long n,
                              unsigned long long index0, const ChirpShape &cs, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (chirp_fits_32(cs)) {
        long long blocks = (n + 1023) / 1024;  // 256 samples per wave
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(chirp_demod_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n,
                           index0, cs);
    } else {
        long long blocks = (n + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(chirp_demod_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in,
                           out, n, index0, cs);
    }
    return hipGetLastError();
}

hipError_t launch_chirp_lockin(const float2 *carry, int carry_len, const float2 *in,
                               const float *profile, int ppt, int valid, float2 *out,
                               unsigned long long index0, const ChirpShape &cs, hipStream_t st,
                               float2 *partial, int partial_cap) {
    if (valid <= 0) return hipSuccess;
    // the fast kernel needs windows made of whole steps that start on a step boundary:
    // true for every window the demodulator forms (ppt = length*decim, see enqueue_chirp)
    const bool fast = chirp_fits_32(cs) && ppt % (int)cs.length == 0 && index0 % cs.length == 0;
    if (fast && partial && valid <= 512) {
        // few points with many samples each: deal a point's stretches to several waves (GSDR_CHIRP_SPLIT=0: never)
        const char *e = std::getenv("GSDR_CHIRP_SPLIT");
        const long long decim = ppt / (long long)cs.length, nst = ((long long)cs.length + 255) / 256, total = decim * nst;
        long long parts = (4096 + valid - 1) / valid;            // ~4096 waves in the launch
        if (parts > total / 4) parts = total / 4;                // at least four stretches per wave
        if (parts > partial_cap / valid) parts = partial_cap / valid;
        if (!(e && e[0] == '0') && parts >= 2 && total >= 16 && total < 0x7fffffffLL) {
            const long long per_part = (total + parts - 1) / parts;
            parts = (total + per_part - 1) / per_part;
            const long long waves = (long long)valid * parts;
            hipLaunchKernelGGL(chirp_lockin_split_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, carry, carry_len, in,
                               profile, ppt, (int)decim, valid, (int)parts, (int)per_part, partial, index0, cs);
            hipLaunchKernelGGL(chirp_lockin_sum_kernel, dim3((unsigned)((valid + 3) / 4)), dim3(256), 0, st, partial, valid,
                               (int)parts, out);
            return hipGetLastError();
        }
    }
    if (fast)
        hipLaunchKernelGGL(chirp_lockin_kernel, dim3((unsigned)((valid + 3) / 4)), dim3(256), 0, st,
                           carry, carry_len, in, profile, ppt, ppt / (int)cs.length, valid, out, index0, cs);
    else
        hipLaunchKernelGGL(chirp_lockin_generic_kernel, dim3((unsigned)((valid + 3) / 4)), dim3(256),
                           0, st, carry, carry_len, in, profile, ppt, valid, out, index0, cs);
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

typedef float gsdr_f2 __attribute__((ext_vector_type(2)));
// one 8-byte store (make_float2 is not always_inline: inside a no-packed-fp32-ops kernel it is a real call)
__device__ __forceinline__ void store_c64(float2 *p, float re, float im) {
    *reinterpret_cast<gsdr_f2 *>(p) = gsdr_f2{re, im};
}

// (Built without packed FP32 and without library calls like every kernel that may meet the matrix-core loop of
// another handle on the GPU -- a TX generator beside the RX demodulators, ref:
// cpp/USRP_server_link_threads.cpp:121,136; DESIGN.md section 4.1, rule 3.  Sines and cosines come from
// sincos_index on a 32-bit fraction of a turn, the logarithm of the Box-Muller radius from v_log_f32.)
__global__ __launch_bounds__(256) GSDR_NO_PK void source_tones_kernel(
    float2 *__restrict__ out, long long n, long long start, unsigned rate,
    const unsigned *__restrict__ fmod, const float *__restrict__ ampl,
    const float *__restrict__ phase, int n_tones, float sigma, unsigned long long seed) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const double inv_rate = 1.0 / (double)rate;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const unsigned long long s = (unsigned long long)(start + j) % rate;
        float re = 0.f, im = 0.f;
        for (int k = 0; k < n_tones; ++k) {
            // exact integer phase (f*s mod rate) plus the tone's initial phase, as a fraction of a turn in
            // double, then one sincos on its leading 32 bits
            const unsigned long long ph = ((unsigned long long)fmod[k] * s) % rate;
            double turns = (double)ph * inv_rate + (double)phase[k] * 0.15915494309189533577;
            turns -= __builtin_floor(turns);
            const unsigned idx = (unsigned)(unsigned long long)(turns * 4294967296.0);
            float sn, cs;
            sincos_index((int)idx, sn, cs);
            const float a = ampl[k];
            re += a * cs;
            im += a * sn;
        }
        if (sigma > 0.f) {
            const unsigned long long h = mix64(seed ^ mix64((unsigned long long)(start + j)));
            const float u1 = ((float)(unsigned)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);
            const float rad = sigma * __builtin_sqrtf(-1.3862943611198906f * __builtin_log2f(u1));   // sqrt(-2 ln u1)
            float sn, cs;
            sincos_index((int)((unsigned)((h >> 8) & 0xffffffu) << 8), sn, cs);      // 24 bits of a turn
            re += rad * cs;
            im += rad * sn;
        }
        store_c64(out + j, re, im);
    }
}

__global__ __launch_bounds__(256) GSDR_NO_PK void source_chirp_kernel(float2 *__restrict__ out, long long n,
                                                           unsigned long long index0,
                                                           ChirpShape cs, float scale) {
    const bool small = cs.period < 0xffffffffull && cs.num_steps < 0xfffffffeull;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += stride) {
        const unsigned long long e = (index0 + (unsigned long long)o) % cs.period;
        float s, c;
        sincos_index(chirp_index(e, cs, small), s, c);
        store_c64(out + o, s * scale, -c * scale);  // ref: kernels.cu:367-368
    }
}

// ---------------------------------------------------------------------------
// TX tone comb at scale (row f3): x[s] = sum_k a_k e^(i phi_k) w_k^s, w_k = e^(+2 pi i f_k / rate).
// ref: tone_gen, cpp/kernels.cu:589-684, builds one period (`rate` samples: 1.6 GB at 200 Msps) by an
// inverse FFT at set-up and TX_buffer_generator::get_from_tones serves slices of it
// (cpp/USRP_buffer_generator.cpp:226-229).  Here a buffer is synthesised when it is asked for, fast
// enough for that (2048 tones: 0.3 ms per 1 Mi samples, 17 x real time at 200 Msps), from exact integer
// phases -- no period in memory, no drift:
//   1024 consecutive samples at a time, s = s0 + 64 j + lane:  w^s = w^s0 * w^lane * w^(64 j);
//   w^s0 comes exactly (integer phase f s0 mod rate, double sincos) once per tone and wave -- 64 tones at a
//   time, one per lane, handed round with v_readlane --, w^lane from a table B[k][64], w^(64 j) from a
//   table C[k][16] through scalar loads: T = (a_k e^(i phi_k) w^s0) B[k][lane], then 16 complex
//   multiply-adds acc[j] += T C[k][j] with scalar second operands.  ~80 vector instructions per tone and
//   1024 samples; the per-sample sincos of source_tones_kernel (bench.py's noisy input) is 70 x that.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) GSDR_NO_PK void tones_synth_kernel(
    float2 *__restrict__ out, long long n, unsigned long long start, unsigned rate, unsigned long long rate_magic,
    double inv_rate, const unsigned *__restrict__ fmod, const float2 *__restrict__ q0,
    const float2 *__restrict__ btab, const float2 *__restrict__ ctab, int n_tones) {
    // a workgroup makes 1024 consecutive samples; its four waves share the tones (chunks of 64 tones go
    // round the waves) and add their partial sums through the LDS: four times the waves of a wave-per-
    // stretch layout, which at one wave per SIMD was a bare latency chain (2048 tones: 708 us per buffer)
    __shared__ float2 part[4][16][64];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long base = (long long)blockIdx.x * 1024;
    const unsigned long long s0 = mod_rate(start + (unsigned long long)base, rate, rate_magic);   // start < rate, base < 2^62
    float ax[16], ay[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) ax[j] = ay[j] = 0.f;
    for (int k0 = wid * 64; k0 < n_tones; k0 += 256) {
        const int kl = k0 + lane < n_tones ? k0 + lane : n_tones - 1;
        const unsigned long long ph = mod_rate((unsigned long long)fmod[kl] * s0, rate, rate_magic);
        double re, im;
        exact_phasor(ph, inv_rate, re, im);                     // e^(-2 pi i ph / rate): the TX sign is the other one
        const float2 a = q0[kl];
        const float wr = (float)re, wi = -(float)im;
        const int Qx = __float_as_int(a.x * wr - a.y * wi), Qy = __float_as_int(a.x * wi + a.y * wr);
        const int cnt = n_tones - k0 < 64 ? n_tones - k0 : 64;
        for (int t = 0; t < cnt; ++t) {
            const float qx = __int_as_float(__builtin_amdgcn_readlane(Qx, t)), qy = __int_as_float(__builtin_amdgcn_readlane(Qy, t));
            const float2 B = btab[(size_t)(k0 + t) * 64 + lane];
            const float tx = qx * B.x - qy * B.y, ty = qx * B.y + qy * B.x;
            const float2 *C = ctab + (size_t)(k0 + t) * 16;      // uniform address: scalar loads
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 c = C[j];
                ax[j] = fmaf(tx, c.x, fmaf(-ty, c.y, ax[j]));
                ay[j] = fmaf(tx, c.y, fmaf(ty, c.x, ay[j]));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) part[wid][j][lane] = mk2c(ax[j], ay[j]);
    __syncthreads();
    // wave w adds the four partial sums of stretches j = 4 w .. 4 w + 3 (in wave order: a fixed summation order)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = wid * 4 + jj;
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float2 v = part[w][j][lane];
            sx += v.x;
            sy += v.y;
        }
        const long long s = base + 64 * j + lane;
        if (s < n) out[s] = mk2c(sx, sy);
    }
}

hipError_t launch_tones_synth(float2 *out, long long n, unsigned long long start, unsigned rate, const unsigned *fmod,
                              const float2 *q0, const float2 *btab, const float2 *ctab, int n_tones, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (!out || rate < 1 || start >= rate || n_tones < 0 || (n_tones > 0 && (!fmod || !q0 || !btab || !ctab)))
        return hipErrorInvalidValue;
    const unsigned long long magic = ~0ULL / rate;
    const long long groups = (n + 1023) / 1024;
    hipLaunchKernelGGL(tones_synth_kernel, dim3((unsigned)groups), dim3(256), 0, st, out, n, start, rate, magic,
                       1.0 / (double)rate, fmod, q0, btab, ctab, n_tones);
    return hipGetLastError();
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

// Does nothing: its first launch makes the runtime load this library's code object and set up
// the stream's queue, so that the first real buffer does not pay for it (gsdr_demod_prepare).
__global__ void warm_kernel() {}

hipError_t launch_warm(hipStream_t st) {
    hipLaunchKernelGGL(warm_kernel, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}

}  // namespace gsdr
