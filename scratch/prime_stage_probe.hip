// probe: the prime-first MFMA stage of csrc/fft_kernels.hip in isolation (scratch only)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define GSDR_NO_PK __attribute__((target("no-packed-fp32-ops")))
namespace gsdr { namespace {
__device__ __forceinline__ float2 mk2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }
__device__ __forceinline__ int fdiv(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }
#include "prime_stage_body.inc"
__global__ __launch_bounds__(1024) GSDR_NO_PK void k(const float2 *x, float2 *y, const float2 *rootsg, int R, int t, int FR, unsigned mag_t) {
    extern __shared__ float2 lds[];
    const int n = R * t, tid = threadIdx.x;
    float2 *src = lds, *dst = lds + FR * n, *roots = dst + FR * n;
    for (int i = tid; i < FR * n; i += 1024) { src[i] = x[i]; dst[i] = mk2(-77.f, -77.f); }
    for (int i = tid; i < R; i += 1024) roots[i] = rootsg[i];
    __syncthreads();
    lds_stage_prime_first_mfma(R, src, dst, n, roots, t, mag_t, FR, tid, 1024);
    __syncthreads();
    for (int i = tid; i < FR * n; i += 1024) y[i] = dst[i];
}
}}
int main() {
    const int cases[][3] = {{17, 1, 1}, {17, 2, 3}, {41, 30, 4}, {127, 8, 4}};
    for (auto &c : cases) {
        const int R = c[0], t = c[1], FR = c[2], n = R * t;
        std::vector<float2> x(FR * n), y(FR * n), roots(R);
        for (int i = 0; i < FR * n; ++i) x[i] = make_float2((float)((i * 7) % 13) - 6.f, (float)((i * 5) % 11) - 5.f);
        for (int m = 0; m < R; ++m) roots[m] = make_float2((float)cos(2 * M_PI * m / R), (float)-sin(2 * M_PI * m / R));
        float2 *dx, *dy, *dr;
        (void)hipMalloc(&dx, x.size() * 8); (void)hipMalloc(&dy, x.size() * 8); (void)hipMalloc(&dr, R * 8);
        (void)hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dr, roots.data(), R * 8, hipMemcpyHostToDevice);
        const unsigned mag_t = t <= 1 ? 0u : (unsigned)(0x100000000ULL / (unsigned long long)t + 1ULL);
        hipLaunchKernelGGL(gsdr::k, dim3(1), dim3(1024), (2 * FR * n + R) * 8, 0, dx, dy, dr, R, t, FR, mag_t);
        (void)hipMemcpy(y.data(), dy, x.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0; int bad = 0, first = -1;
        for (int fr = 0; fr < FR; ++fr) for (int i = 0; i < t; ++i) for (int q = 0; q < R; ++q) {
            double re = 0, im = 0;
            for (int r = 0; r < R; ++r) { const double a = -2 * M_PI * ((q * r) % R) / R; const float2 v = x[fr * n + i + r * t]; re += v.x * cos(a) - v.y * sin(a); im += v.x * sin(a) + v.y * cos(a); }
            const float2 o = y[fr * n + i * R + q];
            const double e = fabs(o.x - re) + fabs(o.y - im);
            if (e > 1e-3) { if (first < 0) first = fr * n + i * R + q; ++bad; }
            worst = e > worst ? e : worst;
        }
        printf("R %d t %d FR %d: worst %.3g, %d wrong outputs, first at %d (q %d) value (%g, %g)\n", R, t, FR, worst, bad, first, first < 0 ? -1 : first % R, first < 0 ? 0. : y[first].x, first < 0 ? 0. : y[first].y);
    }
    return 0;
}
