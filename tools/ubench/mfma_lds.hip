// Microbenchmark (scratch): do ds_read_b128 and v_mfma_f32_32x32x16_f16 overlap on gfx950 when
// hipcc schedules them?  Per iteration: 4 MFMAs (2x2 tile) + R ds_read_b128, operands of iteration
// k+DIST come from the reads of iteration k.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <int R, int PIN>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char raw[];
    lds_byte *lds = (lds_byte *)raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 512) ((__attribute__((address_space(3))) float *)lds)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    // conflict-free addresses: row = lane&31 (stride 272), half = lane>>5
    const int base = wave * 8192 + (lane & 31) * 272 + (lane >> 5) * 16;
    half8 fa[4], fb[4];
    for (int j = 0; j < 4; j++) { fa[j] = *(const __attribute__((address_space(3))) half8 *)(lds + base + j * 32); fb[j] = fa[j]; }
#define STEP(CUR, NXT, IT)                                                                          \
    {                                                                                               \
        _Pragma("unroll") for (int j = 0; j < R; j++)                                               \
            NXT[j & 3] = *(const __attribute__((address_space(3))) half8 *)(lds + base + (((IT) * 4 + j) & 7) * 32 + (j >> 2) * 4352); \
        if (PIN) __builtin_amdgcn_sched_barrier(0);                                                 \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CUR[0], CUR[2], acc[0][0], 0, 0, 0);     \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CUR[0], CUR[3], acc[0][1], 0, 0, 0);     \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CUR[1], CUR[2], acc[1][0], 0, 0, 0);     \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CUR[1], CUR[3], acc[1][1], 0, 0, 0);     \
        if (PIN) __builtin_amdgcn_sched_barrier(0);                                                 \
    }
    for (int it = 0; it < iters; it += 2) {
        STEP(fa, fb, it)
        STEP(fb, fa, it + 1)
    }
    float s = 0;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int i = 0; i < 16; i++) s += acc[a][b][i];
    out[blockIdx.x * 512 + tid] = s;
}

template <int R, int PIN> void run(float *out, int threads, const char *tag)
{
    const int iters = 4000, grid = 256 * (512 / threads) * 2;
    hipFuncSetAttribute((const void *)k<R, PIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<R, PIN><<<grid, threads, 65536>>>(out, iters);
    hipEventRecord(e0);
    k<R, PIN><<<grid, threads, 65536>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * (threads / 64) * iters * 4 * 32768.0;
    printf("%s R=%d pin=%d threads=%d: %.3f ms  %.0f TFLOP/s\n", tag, R, PIN, threads, ms, flops / ms / 1e9);
}

int main()
{
    float *out; hipMalloc(&out, 4 * 512 * 4096);
    run<0, 0>(out, 512, "2w/simd"); run<2, 0>(out, 512, "2w/simd"); run<4, 0>(out, 512, "2w/simd");
    run<4, 1>(out, 512, "2w/simd"); run<6, 1>(out, 512, "2w/simd"); run<8, 1>(out, 512, "2w/simd");
    run<0, 0>(out, 256, "1w/simd"); run<4, 1>(out, 256, "1w/simd"); run<8, 1>(out, 256, "1w/simd");
    return 0;
}
