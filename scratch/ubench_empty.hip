// dispatch cost of a grid of 626 x 256 threads: empty kernel, with small / large register and LDS footprints
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_small(float *o) { if (o == nullptr) o[0] = 1.f; }
__global__ __launch_bounds__(256, 2) void k_big(float *o) {
    __shared__ float lds[8192];
    // force ~240 VGPRs: inline asm clobbers
    asm volatile("" ::: "v100", "v150", "v200", "v240", "a60");
    if (o == nullptr) { lds[threadIdx.x] = 1.f; o[0] = lds[threadIdx.x ^ 1]; }
}
template <typename K> float run(K k, int grid, float *o) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, o);
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 200 * 1e3f;
}
int main() {
    float *o; hipMalloc(&o, 4096);
    for (int grid : {64, 256, 512, 626, 1024, 2048, 8192})
        printf("grid %5d: small %.2f us   big(240 regs, 32 KB LDS) %.2f us\n", grid, run(k_small, grid, o), run(k_big, grid, o));
    return 0;
}
