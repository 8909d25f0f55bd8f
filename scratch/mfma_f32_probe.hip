// probe: layout of v_mfma_f32_32x32x2_f32 and reading its results (scratch only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *D, int K, int nops) {
    const int lane = threadIdx.x, l32 = lane & 31, l2 = lane >> 5;
    f16v acc = {0};
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = A[l32 * K + k0 + l2], b = B[(k0 + l2) * 32 + l32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (nops) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    for (int j = 0; j < 16; ++j) {
        const int row = 8 * (j >> 2) + 4 * l2 + (j & 3);
        D[row * 32 + l32] = acc[j];
    }
}
int main() {
    const int K = 8;
    std::vector<float> A(32 * K), B(K * 32), D(1024), R(1024, 0.f);
    for (int i = 0; i < 32 * K; ++i) A[i] = (float)((i * 7) % 13) - 6.f;
    for (int i = 0; i < K * 32; ++i) B[i] = (float)((i * 5) % 11) - 5.f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int kk = 0; kk < K; ++kk) R[i * 32 + j] += A[i * K + kk] * B[kk * 32 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int nops = 0; nops < 2; ++nops) {
        hipMemset(dD, 0, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, nops);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0, firstbad = -1;
        for (int i = 0; i < 1024; ++i) if (D[i] != R[i]) { if (firstbad < 0) firstbad = i; ++bad; }
        printf("nops %d: %d wrong of 1024, first wrong at row %d col %d\n", nops, bad, firstbad / 32, firstbad % 32);
    }
    return 0;
}
