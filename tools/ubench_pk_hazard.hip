// ubench_pk_hazard.hip -- reproducer for rule R3 of tools/gen_ddc_mfma_ring.py.
//
// Half of the workgroups run a dense MFMA + plain-VALU loop (like the DDC main loop), the
// other half evaluate v_pk_mul_f32 with a HIGH-half broadcast (op_sel:[0,1] op_sel_hi:[1,1])
// and with a LOW-half broadcast (op_sel_hi:[1,0]) and compare every result with v_mul_f32.
// Two workgroups share a CU, i.e. two waves share each SIMD.  Prints the number of wrong
// packed results per form, and the same with the MFMA workgroups replaced by idle ones.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pk_hazard.hip -o /tmp/pkh && /tmp/pkh
#include <hip/hip_runtime.h>

#include <cstdio>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;

__global__ __launch_bounds__(256, 2) void k(unsigned *bad_hi, unsigned *bad_lo, unsigned *lanes_hi, float *sink,
                                            int with_mfma) {
    const int lane = threadIdx.x & 63;
    // workgroup kinds alternate in blocks of 256 so that every CU gets one of each
    const bool mfma_wg = ((blockIdx.x >> 8) & 1) != 0;
    if (mfma_wg) {
        if (!with_mfma) return;
        half8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)(threadIdx.x * 0.001f + i);
            b[i] = (_Float16)(1.0f + i * 0.01f);
        }
        float16v c0 = {0}, c1 = {0};
        float s0 = threadIdx.x, s1 = 0.5f;
        __shared__ uint4 ring[1024];
        ring[threadIdx.x] = make_uint4(1, 2, 3, 4);
        for (int it = 0; it < ITERS; ++it) {
            if ((it & 3) == 0) {
                const uint4 v = ring[(threadIdx.x + it) & 1023];
                ring[(threadIdx.x * 3 + it) & 1023] = v;
                asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0" : "+v"(s0) : "v"(s1));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0" : "+v"(s0) : "v"(s1));
            }
        }
        float r = s0;
        for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
        sink[blockIdx.x * 256 + threadIdx.x] = r;
        return;
    }
    unsigned nh = 0, nl = 0, lh = 0;
    float2v x = {1.0f + lane * 0.25f, 2.0f - lane * 0.125f};
    float2v h = {0.5f + threadIdx.x * 0.001f, 0.25f + threadIdx.x * 0.002f};
    for (int it = 0; it < ITERS * 4; ++it) {
        float2v hi, lo;
        // h is rewritten by a packed multiply right in front of its readers (distance 0 and 1),
        // as the scaled taps were in the DDC conversion that failed
        asm volatile("v_pk_mul_f32 %0, %0, %3\n"
                     "v_pk_mul_f32 %1, %4, %0 op_sel:[0,1] op_sel_hi:[1,1]\n"
                     "v_pk_mul_f32 %2, %4, %0 op_sel_hi:[1,0]"
                     : "+v"(h), "=&v"(hi), "=&v"(lo) : "v"(float2v{1.0001f, 0.9999f}), "v"(x));
        float r0, r1, r2, r3;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(x.x), "v"(h.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(x.y), "v"(h.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r2) : "v"(x.x), "v"(h.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r3) : "v"(x.y), "v"(h.x));
        if (hi.x != r0 || hi.y != r1) {
            nh++;
            lh |= 1u << (lane >> 4);
        }
        if (lo.x != r2 || lo.y != r3) nl++;
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(float2v{0.001f, -0.001f}));
    }
    if (nh) {
        atomicAdd(bad_hi, nh);
        atomicOr(lanes_hi, lh);
    }
    if (nl) atomicAdd(bad_lo, nl);
}

int main() {
    unsigned *d;
    float *sink;
    hipMalloc(&d, 3 * sizeof(unsigned));
    hipMalloc(&sink, 1024 * 256 * sizeof(float));
    for (int with_mfma = 1; with_mfma >= 0; --with_mfma) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0, 3 * sizeof(unsigned));
            hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, d + 1, d + 2, sink, with_mfma);
            hipDeviceSynchronize();
            unsigned h[3];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s  wrong results: high-half broadcast %u (quarter-waves hit: mask 0x%x), low-half broadcast %u  of %llu each\n",
                   with_mfma ? "MFMA loop beside :" : "idle beside      :", h[0], h[2], h[1],
                   256ull * 64 * 4 * ITERS * 4);
        }
    }
    return 0;
}
