// Store-pattern microbenchmark for the DDC epilogue (MI355X): 626 workgroups x 4 waves write a
// [10000][256] float2 output (20.5 MB) as tiles of 32 rows x 128 tones.
//   A  what store_tile() does: per instruction a wave writes 32 tones (256 B) of two rows, 8 B per lane
//   B  the same bytes as 16 B per lane: per instruction a wave writes 128 tones (1 KiB) of one row
//   C  like A with nontemporal stores,  D  like B with nontemporal stores
// hipcc -O3 --offload-arch=gfx950 tools/ubench_store.hip -o /tmp/ubench_store && /tmp/ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float *out, int nout, int N, int ntq, int ngt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int gt = (q / ntq) * 8 + xcd;
    if (gt >= ngt) return;
    const int tq = q % ntq;
    const float v = (float)blockIdx.x;
    if (MODE == 0 || MODE == 2) {
        const int r = lane & 31, hh = lane >> 5;
        const int n = (tq * 4 + wave) * 32 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = gt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (row < nout) {
                float2v y = {v, v + i};
                float2v *p = reinterpret_cast<float2v *>(out) + (size_t)row * N + n;
                if (MODE == 2) __builtin_nontemporal_store(y, p); else *p = y;
            }
        }
    } else {
        // wave w writes rows w*8 .. w*8+7 of the tile, 128 tones (1 KiB) per instruction
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = gt * 32 + wave * 8 + i;
            if (row < nout) {
                float4v y = {v, v + i, v, v};
                float4v *p = reinterpret_cast<float4v *>(out + ((size_t)row * N + tq * 128) * 2) + lane;
                if (MODE == 3) __builtin_nontemporal_store(y, p); else *p = y;
            }
        }
    }
}

template <int MODE>
float run(float *out, int nout, int N, int reps) {
    const int ntq = N / 128, ngt = (nout + 31) / 32, grid = (ngt + 7) / 8 * 8 * ntq;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, nout, N, ntq, ngt);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, nout, N, ntq, ngt);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main() {
    const int cfg[][2] = {{10000, 256}, {1000, 2048}, {40000, 256}};
    for (auto &c : cfg) {
        const int nout = c[0], N = c[1];
        float *out;
        hipMalloc(&out, (size_t)nout * N * 8);
        const double mb = (double)nout * N * 8 / 1e6;
        const float ta = run<0>(out, nout, N, 50), tb = run<1>(out, nout, N, 50), tc = run<2>(out, nout, N, 50),
                    td = run<3>(out, nout, N, 50);
        printf("[%d][%d] %.1f MB:  A 8B/lane %.2f us (%.2f TB/s)   B 16B/lane rows %.2f us (%.2f TB/s)   C nt 8B %.2f us   D nt 16B %.2f us\n",
               nout, N, mb, ta, mb / ta, tb, mb / tb, tc, td);
        hipFree(out);
    }
    return 0;
}
