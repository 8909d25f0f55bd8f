// ubench_read.hip -- the floor of a launch that reads one 1 Mi-sample buffer (8 MB) once:
// what the chirp lock-in (C4) and the staging / absmax pass can be held against.
// Back-to-back launches on one stream over 8 rotating buffers (as bench.py does), timed with events:
//   empty     1250 workgroups that do nothing
//   read8     one wave per 200 samples, 8-byte loads (the lock-in's access pattern), one store per wave
//   read16    the same bytes with 16-byte loads, 4 in flight per lane
//   read16x   grid of 256 / 512 / 1024 workgroups, grid-stride, 16-byte loads, 4 in flight per lane
// hipcc -O3 --offload-arch=gfx950 tools/ubench_read.hip -o /tmp/ubench_read && /tmp/ubench_read
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_empty(const float2 *, float *o, long long) {
    if (o == nullptr) o[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_read8(const float2 *x, float *o, long long n) {
    const int lane = threadIdx.x & 63;
    const long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long base = v * 200;
    if (base >= n) return;
    float2 s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 64 + lane;
        s[i] = x[base + (r < 200 ? r : 0)];
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += s[i].x * s[i].y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) o[v] = acc;
}

__global__ __launch_bounds__(256) void k_read16(const float2 *x, float *o, long long n) {
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const long long n4 = n / 2, stride = (long long)gridDim.x * 256;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        float4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = i + k * stride < n4 ? x4[i + k * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += q[k].x * q[k].y + q[k].z * q[k].w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) o[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

// the staging pass's tail: the workgroup's maximum into one of `slots` addresses with atomicMax
// (slots = 0: a plain store of the partial, one address per workgroup)
template <int SLOTS>
__global__ __launch_bounds__(256) void k_read16_max(const float2 *x, float *o, long long n) {
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const long long n4 = n / 2, stride = (long long)gridDim.x * 256;
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        float4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = i + k * stride < n4 ? x4[i + k * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned a = __float_as_uint(q[k].x) & 0x7fffffffu, b = __float_as_uint(q[k].y) & 0x7fffffffu;
            const unsigned c = __float_as_uint(q[k].z) & 0x7fffffffu, d = __float_as_uint(q[k].w) & 0x7fffffffu;
            const unsigned u = a > b ? a : b, w = c > d ? c : d;
            m = m > u ? m : u;
            m = m > w ? m : w;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, off, 64);
        m = m > t ? m : t;
    }
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned u = wmax[0] > wmax[1] ? wmax[0] : wmax[1], w = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
        const unsigned t = u > w ? u : w;
        unsigned *slots = reinterpret_cast<unsigned *>(o);
        if (SLOTS > 0)
            atomicMax(&slots[blockIdx.x % (SLOTS > 0 ? SLOTS : 1)], t);
        else
            slots[blockIdx.x] = t;
    }
}

template <typename K>
float run(K k, int grid, float2 **bufs, float *o, long long n) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, bufs[i % 8], o, n);
    hipEventRecord(a);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, bufs[i % 8], o, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 2000 * 1e3f;
}

int main() {
    const long long n = 1000000;
    float2 *bufs[8];
    for (auto &b : bufs) {
        hipMalloc(&b, n * sizeof(float2));
        hipMemset(b, 0x11, n * sizeof(float2));
    }
    float *o;
    hipMalloc(&o, 1 << 20);
    const double mb = n * 8 / 1e6;
    float t;
    t = run(k_empty, 1250, bufs, o, n);
    printf("empty    grid 1250: %.2f us per launch\n", t);
    t = run(k_read8, 1250, bufs, o, n);
    printf("read8    grid 1250: %.2f us per launch  %.2f TB/s\n", t, mb / t);
    for (int grid : {122, 245, 489, 977}) {       // 8, 4, 2, 1 rounds of 4 x 16-byte loads per lane
        t = run(k_read16, grid, bufs, o, n);
        printf("read16   grid %4d: %.2f us per launch  %.2f TB/s\n", grid, t, mb / t);
    }
    t = run(k_read16_max<16>, 245, bufs, o, n);
    printf("read16 + max, 16 atomic addresses, grid 245: %.2f us per launch\n", t);
    t = run(k_read16_max<64>, 245, bufs, o, n);
    printf("read16 + max, 64 atomic addresses, grid 245: %.2f us per launch\n", t);
    t = run(k_read16_max<0>, 245, bufs, o, n);
    printf("read16 + max, plain store per workgroup, grid 245: %.2f us per launch\n", t);
    t = run(k_read16_max<0>, 489, bufs, o, n);
    printf("read16 + max, plain store per workgroup, grid 489: %.2f us per launch\n", t);
    t = run(k_read16_max<0>, 977, bufs, o, n);
    printf("read16 + max, plain store per workgroup, grid 977: %.2f us per launch\n", t);
    return 0;
}
