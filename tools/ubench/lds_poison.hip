// Scratch (GPU): fill the LDS of every CU with a pattern, so that a kernel reading LDS bytes it never
// wrote shows a pattern-dependent result.  hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/lds_poison.hip -o tools/ubench/liblds_poison.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512) void k_poison(unsigned pattern)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 512) lds[i] = pattern;
    __syncthreads();
    if (lds[threadIdx.x] != pattern) __builtin_trap();
}
extern "C" int lds_poison(void *stream, unsigned pattern)
{
    hipFuncSetAttribute((const void *)k_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_poison, dim3(2048), dim3(512), 160 * 1024, (hipStream_t)stream, pattern);
    return (int)hipGetLastError();
}
