// tower_common.hpp -- fused residual trunk of the policy/value tower for gfx950 (MFMA, fp16 in,
// fp32 accumulate): the design, and the types, constants and helpers the trunk kernels share.
// The production kernels are tower_x16.hpp (v_mfma_f32_16x16x32_f16; 64, 128 and 256 filters).
// (The round-1 kernels on v_mfma_f32_32x32x16_f16 and their tuning harness are in the git history up to round 5:
// tools/ubench/r1_kernels/, trunk_variants.hip.)
//
// Replaces the trunk of ChessModel (/root/reference/src/chessrl/model.py:33-37,111-122: stem
// Conv3x3 + N x [Conv3x3-BN-ReLU-Conv3x3-BN-add-ReLU]) for inference.  BatchNorm is folded into
// the convolution weights/bias on the host (chessrl_amd/model.py).
//
// MI355X design.  A chess board is 8x8 = 64 positions: with NHWC activations a whole board is a
// 64 x F fp16 tile, so a workgroup keeps the activations of its boards (4 at 64 / 128 filters, 2
// at 256) resident in LDS for the ENTIRE tower: no activation ever goes back to HBM between
// layers.  Per workgroup:
//   * 8 waves (2 per SIMD); a wave owns 64 (or 32) positions x 64 (or 32) output channels: a fp32
//     accumulator tile plus the fp32 RESIDUAL STREAM of the same tile in registers -- the skip
//     connection never leaves registers and is never rounded to fp16;
//   * the 3x3 convolution is an implicit GEMM over 9 taps x Cin channels.  The activation
//     operand of tap (dy,dx) is read straight from the LDS-resident board at row p + 8dy + dx;
//     off-board neighbours read a row of zeros with the same bank residue (one v_cndmask on the
//     ADDRESS per read);
//   * weights stream global -> LDS with global_load_lds (16 B/lane, no VGPR round trip) as
//     16-KiB (8-KiB at 64 filters) tiles through a ring of LDS slots, several tiles in flight,
//     counted s_waitcnt vmcnt + raw s_barrier; the stream runs across layer boundaries (weights
//     do not depend on activations), so the MFMA pipe only drains at the single extra barrier per
//     layer that separates the last read of the activation buffer from the epilogue's in-place
//     rewrite;
//   * the product is computed transposed (D[out-ch][pos] = W . X^T): a lane then holds 4
//     CONSECUTIVE channels of one position per accumulator quad, so the epilogue (residual add,
//     ReLU, fp16 pack) writes whole words back into the NHWC LDS image;
//   * LDS images are padded / swizzled in 16-B chunks so that every ds_read_b128 lane group covers
//     all 64 banks (tower_x16.hpp: Geo16).
// The three 1x1 head convolutions are reduced in the kernel's tail; the dense layers are
// heads.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace crl_tower {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BOARDS_PER_WG = 4;              // n_boards handed to crl_trunk_forward is a multiple of this
constexpr int CH = 128;                       // input planes (127 + one zero pad)
constexpr int ROW_BYTES = CH * 2;             // one position of the fp16 input planes
constexpr int BOARD_BYTES = 64 * ROW_BYTES;   // 16 KiB
constexpr int MAX_CONVS = 41;                 // stem + 2 * 20 blocks
constexpr int PIPE_RING = 4;                  // weight tiles in the plain LDS ring
constexpr int LIST_HEADER = 4;                // int32 words in front of a board list (CRL_LIST_HEADER, include/chessrl_hip.h)

typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ inline half8 lds_read16(const lds_byte *base, int off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) half8 *>(base + off);
}

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ds_read_b128 the compiler does not see.  hipcc undoes a hand-placed software pipeline in two
// ways: its scheduler sinks LDS reads to their first use, and its waitcnt pass answers a pinned
// prefetch with lgkmcnt(0) right behind the newest reads.  So fragment reads are inline asm,
// counted by hand (wait_lgkm<N>), and the order is pinned with sched_barrier(0).
template <int OFF>
__device__ __forceinline__ half8 lds_read16_asm(int addr)
{
    half8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

// Input given as 128 plane bitboards per board (u64 [n_boards][128]; bit sq of plane c = channel c
// on square sq -- what the encoder builds before it would expand them, csrc/search.hpp): expand them
// into the padded fp16 activation rows of the NB resident boards.  One item = (board, position,
// 16 channels) = two 16-byte LDS stores; NB items per thread of a 512-thread workgroup.
template <int NB, int AROW, int ABOARD>
__device__ inline void expand_bitplanes(const unsigned char *planes, lds_byte *lds, size_t wg_board0, int tid)
{
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(planes) + wg_board0 * 128;
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int item = k * 512 + tid;
        const int b = item >> 9, p = (item >> 3) & 63, c = item & 7;
        const int sq = p ^ 56;                          // row 0 of the planes is rank 8
        const unsigned long long *m = src + b * 128 + c * 16;
        unsigned int w[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            w[q] = (((m[2 * q] >> sq) & 1) ? 0x3C00u : 0u) | (((m[2 * q + 1] >> sq) & 1) ? 0x3C000000u : 0u);
        lds_byte *dst = lds + b * ABOARD + p * AROW + c * 32;
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(dst) = u32x4{w[0], w[1], w[2], w[3]};
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(dst + 16) = u32x4{w[4], w[5], w[6], w[7]};
    }
}

// the same for boards given one by one (the indexed launch: resident board k is row rows[k] of `planes`)
template <int NB, int AROW, int ABOARD>
__device__ inline void expand_bitplanes_rows(const unsigned char *planes, lds_byte *lds, const int (&rows)[NB], int tid)
{
    const unsigned long long *all = reinterpret_cast<const unsigned long long *>(planes);
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int item = k * 512 + tid;
        const int b = item >> 9, p = (item >> 3) & 63, c = item & 7;     // b == k: one board per 512 items
        const int sq = p ^ 56;
        const unsigned long long *m = all + (size_t)rows[k] * 128 + c * 16;
        unsigned int w[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            w[q] = (((m[2 * q] >> sq) & 1) ? 0x3C00u : 0u) | (((m[2 * q + 1] >> sq) & 1) ? 0x3C000000u : 0u);
        lds_byte *dst = lds + b * ABOARD + p * AROW + c * 32;
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(dst) = u32x4{w[0], w[1], w[2], w[3]};
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(dst + 16) = u32x4{w[4], w[5], w[6], w[7]};
    }
}

}  // namespace crl_tower
