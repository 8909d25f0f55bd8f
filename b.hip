#include <hip/hip_runtime.h>
__global__ void k(float*a,float*b){ a[threadIdx.x]=__builtin_amdgcn_fmul_legacy(a[threadIdx.x],b[threadIdx.x]); }
