// ubench.hip -- instruction-rate microbenchmarks that decide the DDC kernel design
// on gfx950 (results are quoted in DESIGN.md).  Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                  \
            return 1;                                                             \
        }                                                                         \
    } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int ITERS = 2048;

// 16 independent v_fmac_f32 with an SGPR multiplicand per iteration
__global__ __launch_bounds__(256) void k_fma_sgpr(float *out, float s) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    float b = threadIdx.x * 1e-6f + 1.0f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(b));
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// 16 independent v_pk_fma_f32 (2 fma each) per iteration
__global__ __launch_bounds__(256) void k_pk_fma(float *out, float s) {
    float2v a[16];
    for (int i = 0; i < 16; ++i) a[i] = float2v{threadIdx.x * 0.001f + i, 1.0f * i};
    float2v b = float2v{threadIdx.x * 1e-6f + 1.0f, 0.5f};
    float2v c = float2v{s, s};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(b));
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// v_pk_fma_f32 with an SGPR pair as one source (what a packed FIR MAC would use)
__global__ __launch_bounds__(256) void k_pk_fma_sgpr(float *out, float s) {
    float2v a[16];
    for (int i = 0; i < 16; ++i) a[i] = float2v{threadIdx.x * 0.001f + i, 1.0f * i};
    float2v b = float2v{threadIdx.x * 1e-6f + 1.0f, 0.5f};
    float2v c = float2v{s, s};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(c), "v"(b));
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// 8 independent v_mfma_f32_4x4x1_16b_f32 per iteration (512 flop each)
__global__ __launch_bounds__(256) void k_mfma4(float *out, float s) {
    float4v acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float4v{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f + s, b = threadIdx.x * 1e-4f + 1.0f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// the hybrid DDC step in one wave: per "sample" 4 VALU (mix) + 2 MFMA 4x4x1 (FIR)
__global__ __launch_bounds__(256) void k_hybrid(float *out, float s) {
    float4v ar[2], ai[2];
    for (int i = 0; i < 2; ++i) ar[i] = ai[i] = float4v{0, 0, 0, 0};
    float br = threadIdx.x * 1e-3f + 1.0f, bi = threadIdx.x * 1e-4f + 0.5f;
    float taps = threadIdx.x * 1e-5f + 0.1f;
    float xr = s, xi = s * 0.5f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float ur, ui;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ur) : "s"(xi), "v"(bi));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ui) : "s"(xi), "v"(br));
            asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(ur) : "s"(xr), "v"(br));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ui) : "s"(xr), "v"(bi));
            ar[k & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(taps, ur, ar[k & 1], 0, 0, 0);
            ai[k & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(taps, ui, ai[k & 1], 0, 0, 0);
        }
    }
    float r = 0;
    for (int i = 0; i < 2; ++i) r += ar[i].x + ar[i].y + ar[i].z + ar[i].w + ai[i].x + ai[i].y + ai[i].z + ai[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// the all-VALU DDC step: per sample 4 (mix) + 8 (FIR, F=4) VALU with SGPR operands
__global__ __launch_bounds__(256) void k_valu_ddc(float *out, float s) {
    float sr[4], si[4];
    for (int i = 0; i < 4; ++i) sr[i] = si[i] = 0.f;
    float br = threadIdx.x * 1e-3f + 1.0f, bi = threadIdx.x * 1e-4f + 0.5f;
    float xr = s, xi = s * 0.5f, h0 = s * 0.1f, h1 = s * 0.2f, h2 = s * 0.3f, h3 = s * 0.4f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float ur, ui;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ur) : "s"(xi), "v"(bi));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ui) : "s"(xi), "v"(br));
            asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(ur) : "s"(xr), "v"(br));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ui) : "s"(xr), "v"(bi));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sr[0]) : "s"(h0), "v"(ur));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(si[0]) : "s"(h0), "v"(ui));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sr[1]) : "s"(h1), "v"(ur));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(si[1]) : "s"(h1), "v"(ui));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sr[2]) : "s"(h2), "v"(ur));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(si[2]) : "s"(h2), "v"(ui));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sr[3]) : "s"(h3), "v"(ur));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(si[3]) : "s"(h3), "v"(ui));
        }
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) r += sr[i] + si[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// packed DDC step, as ddc_flat_kernel issues it: per sample 2 + 4 v_pk instructions
__device__ __forceinline__ void pk_ddc_body(float2v &s0, float2v &s1, float2v &s2, float2v &s3, float2v b, float2v x, float2v h01, float2v h23) {
    float2v t, u;
    asm volatile(
        "v_pk_mul_f32 %[t], %[x], %[b] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
        "v_pk_fma_f32 %[u], %[x], %[b], %[t] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[s0], %[h01], %[u], %[s0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[s1], %[h01], %[u], %[s1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %[s2], %[h23], %[u], %[s2] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[s3], %[h23], %[u], %[s3] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3), [t] "=&v"(t), [u] "=&v"(u)
        : [b] "v"(b), [x] "s"(x), [h01] "s"(h01), [h23] "s"(h23));
}

// mode 0: every wave runs the packed VALU DDC body; mode 1: every wave runs f32 MFMA 32x32x2;
// mode 2: even workgroups run the VALU body, odd ones the MFMA loop (do the two pipes of a SIMD overlap?)
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_split(float *out, float s, int mode) {
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool do_mfma = mode == 1 || (mode == 2 && ((blockIdx.x >> 8) & 1));  // rounds of 256 workgroups alternate: with round-robin placement every SIMD holds both kinds
    float r = 0;
    if (!do_mfma) {
        float2v s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0};
        float2v b = {threadIdx.x * 1e-3f + 1.0f, threadIdx.x * 1e-4f + 0.5f};
        float2v x = {s, s * 0.5f}, h01 = {s * 0.1f, s * 0.2f}, h23 = {s * 0.3f, s * 0.4f};
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) pk_ddc_body(s0, s1, s2, s3, b, x, h01, h23);
        }
        r = s0.x + s0.y + s1.x + s1.y + s2.x + s2.y + s3.x + s3.y;
    } else {
        float16v acc[2];
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        float a = threadIdx.x * 1e-3f + s, b = threadIdx.x * 1e-4f + 1.0f;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 1], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 16; ++e) r += acc[i][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// scalar-load stream: each wave walks a buffer with s_load_dwordx16 and folds it into a VGPR
__global__ __launch_bounds__(256) void k_sload(float *out, const float *__restrict__ buf, int words) {
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float *p = buf + ((blockIdx.x * 4 + wid) * 1024) % (words - 8192);  // q reaches p + 4111
    float acc = threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
        const float *q = p + (it * 16) % 4096;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = fmaf(q[i], acc, 1.0f);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename K, typename... A>
static double run(const char *name, double ops_per_thread_iter_flop, int blocks, K kern, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double sec = ms * 1e-3 / reps;
    const double flop = ops_per_thread_iter_flop * (double)ITERS * 256.0 * blocks;
    printf("%-14s blocks=%5d  %8.3f ms  %8.2f TFLOP/s\n", name, blocks, sec * 1e3, flop / sec / 1e12);
    fflush(stdout);
    return sec;
}

int main() {
    float *out, *buf;
    CHECK(hipMalloc((void **)&out, sizeof(float) * 256 * 8192));
    CHECK(hipMalloc((void **)&buf, sizeof(float) * (1 << 20)));
    CHECK(hipMemset(buf, 0, sizeof(float) * (1 << 20)));
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s, %d CUs, clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    for (int occ : {1, 2, 4, 8}) {
        const int blocks = pr.multiProcessorCount * occ;  // occ waves per SIMD
        printf("--- %d wave(s) per SIMD\n", occ);
        run("fma_sgpr", 16 * 2.0, blocks, k_fma_sgpr, out, 1.0001f);
        run("pk_fma", 16 * 4.0, blocks, k_pk_fma, out, 1.0001f);
        run("pk_fma_sgpr", 16 * 4.0, blocks, k_pk_fma_sgpr, out, 1.0001f);
        run("mfma4x4x1", 8 * 512.0 / 64.0, blocks, k_mfma4, out, 1.0001f);
        // per sample: 22 "algorithmic" flops (6 mix + 16 FIR) -> report tone-samples instead
        double t = run("hybrid", 8 * 22.0, blocks, k_hybrid, out, 1.0001f);
        printf("   hybrid   : %.3f T tone-samples/s\n", 8.0 * ITERS * 256.0 * blocks / t / 1e12);
        t = run("valu_ddc", 8 * 22.0, blocks, k_valu_ddc, out, 1.0001f);
        printf("   valu_ddc : %.3f T tone-samples/s\n", 8.0 * ITERS * 256.0 * blocks / t / 1e12);
        // k_split: VALU waves do 8 samples/iter (48 pk instr); MFMA waves 4 x 32x32x2 (4096 flop each) per iter
        t = run("split_valu", 8 * 22.0, blocks, k_split, out, 1.0001f, 0);
        printf("   all-VALU : %.3f T tone-samples/s\n", 8.0 * ITERS * 256.0 * blocks / t / 1e12);
        t = run("split_mfma", 4 * 4096.0 / 64.0, blocks, k_split, out, 1.0001f, 1);
        printf("   all-MFMA : %.2f TFLOP/s\n", 4 * 4096.0 / 64.0 * ITERS * 256.0 * blocks / t / 1e12);
        t = run("split_both", 0.5 * 8 * 22.0 + 0.5 * 4 * 4096.0 / 64.0, blocks, k_split, out, 1.0001f, 2);
        printf("   half/half: VALU half %.3f T tone-samples/s + MFMA half %.2f TFLOP/s in %.3f ms\n",
               0.5 * 8.0 * ITERS * 256.0 * blocks / t / 1e12, 0.5 * 4 * 4096.0 / 64.0 * ITERS * 256.0 * blocks / t / 1e12, t * 1e3);
        t = run("sload_x16", 16 * 2.0, blocks, k_sload, out, (const float *)buf, 1 << 20);
        printf("   sload    : %.2f TB/s scalar bytes (64 B per wave per 16 fma)\n",
               64.0 * ITERS * 4.0 * blocks / t / 1e12);
    }
    hipFree(out);
    hipFree(buf);
    return 0;
}
