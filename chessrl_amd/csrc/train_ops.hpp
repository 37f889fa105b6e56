// train_ops.hpp -- the two data-movement kernels of the training step.
//
// On an 8x8 board a 3x3 'same' convolution over NHWC activations is the GEMM
// [B*64, 9*C] x [9*C, Cout]; chessrl_amd/train.py runs that GEMM (and its two backward GEMMs) on
// rocBLAS/hipBLASLt for any batch size and needs only the patch matrix and its adjoint:
//   k_im2col3x3   cols[b][p][tap][c] = x[b][p + d(tap)][c]   (0 outside the board)
//   k_col2im3x3   gx[b][p][c] = sum_tap gcols[b][p - d(tap)][tap][c]   (gather form: no atomics,
//                 fixed summation order tap = 0..8, so the backward pass is deterministic)
// tap = ky*3 + kx, d(tap) = (ky-1, kx-1); rows y, columns x of the NHWC [B][8][8][C] tensor.
// Both are pure HBM streams (10 and 10 floats moved per input float): float4 per thread, the channel
// index fastest so that a wavefront reads and writes contiguous 1-KiB runs.
#pragma once
#include <hip/hip_runtime.h>

namespace crl_train {

__global__ __launch_bounds__(256) void k_im2col3x3(const float4 *__restrict__ x, float4 *__restrict__ cols,
                                                    long long total, int c4)
{
    // one thread per float4 of cols: index = ((b*64 + p)*9 + tap)*c4 + c
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4);
    const long long t = i / c4;
    const int tap = (int)(t % 9);
    const long long bp = t / 9;
    const int p = (int)(bp & 63);
    const int yy = (p >> 3) + tap / 3 - 1, xx = (p & 7) + tap % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)yy < 8u && (unsigned)xx < 8u) v = x[((bp & ~63ll) + yy * 8 + xx) * c4 + c];
    cols[i] = v;
}

__global__ __launch_bounds__(256) void k_col2im3x3(const float4 *__restrict__ gcols, float4 *__restrict__ gx,
                                                    long long total, int c4)
{
    // one thread per float4 of gx: index = (b*64 + p)*c4 + c
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4);
    const long long bp = i / c4;
    const int p = (int)(bp & 63);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int yy = (p >> 3) - (tap / 3 - 1), xx = (p & 7) - (tap % 3 - 1);   // the row that read us
        if ((unsigned)yy < 8u && (unsigned)xx < 8u) {
            const float4 v = gcols[(((bp & ~63ll) + yy * 8 + xx) * 9 + tap) * c4 + c];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    gx[i] = acc;
}

}  // namespace crl_train
