// ddc_device.h -- small device helpers shared by the DDC kernels (internal).
#pragma once
#include <hip/hip_runtime.h>

#include "ddc_kernels.h"

namespace gsdr {

// First block of chunk c when nblk blocks are split into nch chunks whose
// lengths differ by at most one (the first `crem` chunks are one block longer).
// No division: cbase = nblk / nch and crem = nblk % nch come from the host.
__device__ __forceinline__ int chunk_begin(int c, const DdcShape &sh) {
    return c * sh.cbase + (c < sh.crem ? c : sh.crem);
}

// x mod rate for x < 2^63 with magic = floor((2^64 - 1) / rate) (Barrett): the
// quotient estimate is at most 2 short, so at most two subtractions follow.
__device__ __forceinline__ unsigned long long mod_rate(unsigned long long x, unsigned rate,
                                                       unsigned long long magic) {
    const unsigned long long q = __umul64hi(x, magic);
    unsigned long long r = x - q * rate;
    if (r >= rate) r -= rate;
    if (r >= rate) r -= rate;
    return r;
}

// exp(-2*pi*i * ph/rate) for an exact integer phase 0 <= ph < rate, in double to
// ~1e-13: octant reduction on t = ph/rate (t - k/8 is exact), Taylor series of
// degree 11/10 on |theta| <= pi/8, then a rotation by k*pi/4.
__device__ __forceinline__ void exact_phasor(unsigned long long ph, double inv_rate, double &re,
                                             double &im) {
    const double t = (double)ph * inv_rate;            // turns, [0, 1]
    const int k = (int)(t * 8.0 + 0.5);                // nearest eighth of a turn, 0..8
    const double th = (t - 0.125 * (double)k) * 6.283185307179586476925;   // |th| <= pi/8
    const double z = th * th;
    double s = -2.5052108385441718775e-08;             // -1/11!
    s = fma(s, z, 2.7557319223985890653e-06);          //  1/9!
    s = fma(s, z, -1.9841269841269841270e-04);         // -1/7!
    s = fma(s, z, 8.3333333333333333333e-03);          //  1/5!
    s = fma(s, z, -1.6666666666666666667e-01);         // -1/3!
    s = fma(s * z, th, th);
    double c = -2.7557319223985890653e-07;             // -1/10!
    c = fma(c, z, 2.4801587301587301587e-05);          //  1/8!
    c = fma(c, z, -1.3888888888888888889e-03);         // -1/6!
    c = fma(c, z, 4.1666666666666666667e-02);          //  1/4!
    c = fma(c, z, -0.5);
    c = fma(c, z, 1.0);
    // rotate by k * pi/4: (ck, sk) in {0, +-1, +-sqrt(1/2)}
    const double h = 0.70710678118654752440;
    const int kk = k & 7;
    const double ck = (kk == 0) ? 1.0 : (kk == 1 || kk == 7) ? h : (kk == 2 || kk == 6) ? 0.0
                    : (kk == 4) ? -1.0 : -h;
    const double sk = (kk == 0 || kk == 4) ? 0.0 : (kk == 1 || kk == 3) ? h : (kk == 2) ? 1.0
                    : (kk == 6) ? -1.0 : -h;
    const double cs = c * ck - s * sk;                 // cos(2*pi*t)
    const double sn = s * ck + c * sk;                 // sin(2*pi*t)
    re = cs;
    im = -sn;
}

}  // namespace gsdr
