// ubench_mfma.hip -- how many vector instructions of which kind fit in the shadow of
// v_mfma_f32_32x32x16_f16 on gfx950, with one or two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o /tmp/ubm && /tmp/ubm
// Prints shader cycles per MFMA (s_memtime) for: MFMA alone; MFMA + N fillers per gap.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 512;

#define FILL_NONE ""
#define FILL_PKFMA "v_pk_fma_f32 %[p0], %[q], %[p0], %[p0]\n"
#define FILL_PKMUL "v_pk_mul_f32 %[p0], %[q], %[p0]\n"
#define FILL_FMA "v_fma_f32 %[s0], %[s1], %[s0], %[s0]\n"
#define FILL_CVT "v_cvt_pk_f16_f32 %[s2], %[s0], %[s1]\n"
#define FILL_MIX "v_fma_mix_f32 %[s0], %[s0], 1.0, -%[s2] op_sel_hi:[0,0,1]\n"
#define FILL_MOV "v_mov_b32 %[s2], %[s0]\n"

template <int KIND, int N>
__global__ __launch_bounds__(256) void k(long long *cycles, float *sink) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(threadIdx.x * 0.001f + i);
        b[i] = (_Float16)(1.0f + i * 0.01f);
    }
    float16v c0 = {0}, c1 = {0};
    float2v p0 = {1.0f, 2.0f}, q = {0.999f, 1.001f};
    float s0 = threadIdx.x, s1 = 0.5f, s2 = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#define STEP(C)                                                                                   \
    asm volatile("v_mfma_f32_32x32x16_f16 %[c], %[a], %[b], %[c]\n"                              \
                 : [c] "+v"(C) : [a] "v"(a), [b] "v"(b));                                         \
    for (int n = 0; n < N; ++n) {                                                                 \
        if (KIND == 1) asm volatile(FILL_PKFMA : [p0] "+v"(p0) : [q] "v"(q));                     \
        if (KIND == 2) asm volatile(FILL_PKMUL : [p0] "+v"(p0) : [q] "v"(q));                     \
        if (KIND == 3) asm volatile(FILL_FMA : [s0] "+v"(s0) : [s1] "v"(s1));                     \
        if (KIND == 4) asm volatile(FILL_CVT : [s2] "+v"(s2) : [s0] "v"(s0), [s1] "v"(s1));       \
        if (KIND == 5) asm volatile(FILL_MIX : [s0] "+v"(s0) : [s2] "v"(s2));                     \
        if (KIND == 6) asm volatile(FILL_MOV : [s2] "+v"(s2) : [s0] "v"(s0));                     \
    }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            STEP(c0)
            STEP(c1)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    float r = s0 + s2 + p0.x + p0.y;
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND, int N>
static void run(const char *name, int wgs_per_cu, long long *d_cyc, float *d_sink) {
    const int blocks = 256 * wgs_per_cu;
    std::vector<long long> h(blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (long long v : h) avg += (double)v;
    avg /= blocks;
    const double mfmas = ITERS * 8.0;
    // s_memtime runs at a fixed 100 MHz-class clock on some parts: report wall time too
    printf("%-10s N=%d waves/SIMD=%d  counter ticks/MFMA %.2f   wall ns per MFMA per SIMD %.2f\n", name, N, wgs_per_cu,
           avg / mfmas, ms * 1e6 / (mfmas * wgs_per_cu));
    fflush(stdout);
}

int main() {
    long long *d_cyc;
    float *d_sink;
    hipMalloc(&d_cyc, 1024 * sizeof(long long));
    hipMalloc(&d_sink, 1024 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("none", w, d_cyc, d_sink);
        run<1, 2>("pk_fma", w, d_cyc, d_sink);
        run<1, 4>("pk_fma", w, d_cyc, d_sink);
        run<1, 6>("pk_fma", w, d_cyc, d_sink);
        run<2, 4>("pk_mul", w, d_cyc, d_sink);
        run<3, 4>("fma", w, d_cyc, d_sink);
        run<3, 6>("fma", w, d_cyc, d_sink);
        run<4, 4>("cvt_pk", w, d_cyc, d_sink);
        run<5, 4>("fma_mix", w, d_cyc, d_sink);
        run<6, 4>("mov", w, d_cyc, d_sink);
        run<6, 6>("mov", w, d_cyc, d_sink);
    }
    return 0;
}
