#include <hip/hip_runtime.h>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void k(const float2v *in, half2v *hi, half2v *lo, float s) {
    float2v v = in[threadIdx.x] * s;
    half2v h = __builtin_convertvector(v, half2v);
    float2v r = v - __builtin_convertvector(h, float2v);
    hi[threadIdx.x] = h;
    lo[threadIdx.x] = __builtin_convertvector(r, half2v);
}
__global__ void m(const half8 *a, const half8 *b, float16v *c) {
    float16v z = {0};
    float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x+64], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
