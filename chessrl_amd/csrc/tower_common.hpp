// tower_common.hpp -- fused residual trunk of the policy/value tower for gfx950 (MFMA, fp16 in,
// fp32 accumulate): the design, and the types and constants every trunk kernel shares.  The
// production kernels are tower_pipe.hpp (128 filters) and tower_gen.hpp (64 / 256 filters, small
// batches).  tower.hpp, the first build of this design and the baseline of the tuning ladder, is
// compiled only into the tuning library (-DCRL_TUNING), never into libchessrl_hip.so.
//
// Replaces the trunk of ChessModel (/root/reference/src/chessrl/model.py:33-37,111-122: stem
// Conv3x3 + N x [Conv3x3-BN-ReLU-Conv3x3-BN-add-ReLU]) for inference.  BatchNorm is folded into
// the convolution weights/bias on the host (chessrl_amd/model.py).
//
// MI355X design.  A chess board is 8x8 = 64 positions: with NHWC activations a whole board is a
// 64 x 128 fp16 tile of 16 KiB, so a workgroup keeps the activations of its 4 boards (M = 256
// GEMM rows) resident in LDS for the ENTIRE tower: no activation ever goes back to HBM between
// layers.  Per workgroup:
//   * 8 waves (2 per SIMD); wave w owns board w/2 and output channels [64*(w&1), +64): a
//     64(pos) x 64(ch) fp32 accumulator tile = 2x2 MFMA 32x32x16 tiles (64 VGPRs), plus the
//     fp32 RESIDUAL STREAM of the same tile in another 64 VGPRs -- the skip connection never
//     leaves registers and is never rounded to fp16;
//   * the 3x3 convolution is an implicit GEMM over 9 taps x 128 channels.  The activation
//     operand of tap (dy,dx) is read straight from the LDS-resident board at row p + 8dy + dx;
//     off-board neighbours read a 256-byte row of zeros (one v_cndmask on the ADDRESS per read);
//   * weights stream global -> LDS with global_load_lds (16 B/lane, no VGPR round trip) as
//     16-KiB tiles [128 out-ch][64 in-ch] through a 3-buffer ring, two tiles in flight; ONE
//     raw s_barrier per K-step with a counted s_waitcnt vmcnt(2); the stream runs across layer
//     boundaries (weights do not depend on activations), so the MFMA pipe only drains at the
//     single extra barrier per layer that separates the last read of the activation buffer
//     from the epilogue's in-place rewrite;
//   * the product is computed transposed (D[out-ch][pos] = W . X^T): a lane then holds 4
//     CONSECUTIVE channels of one position per accumulator quad, so the epilogue (bias, residual
//     add, ReLU, fp16 pack) writes 8-byte words back into the NHWC LDS image;
//   * LDS images are XOR-swizzled in 16-B chunks (activations: chunk ^ (pos & 15); weight tile:
//     chunk ^ ((row >> 1) & 7), applied on the SOURCE address of the LDS-DMA) so that every
//     ds_read_b128 lane group covers all 64 banks.
// The three 1x1 head convolutions are reduced in the kernel's tail; the dense layers (< 1 % of the
// FLOPs) stay in PyTorch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crl_tower {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BOARDS_PER_WG = 4;
constexpr int CH = 128;                       // channels in and out of every trunk conv
constexpr int ROW_BYTES = CH * 2;             // one position, fp16
constexpr int BOARD_BYTES = 64 * ROW_BYTES;   // 16 KiB
constexpr int ACT_BYTES = BOARDS_PER_WG * BOARD_BYTES;   // 64 KiB
constexpr int ZERO_OFF = ACT_BYTES;           // 256 B of zeros
constexpr int MAX_CONVS = 41;                 // stem + 2 * 20 blocks
constexpr int BIAS_OFF = ZERO_OFF + 256;      // float [MAX_CONVS][128]
constexpr int WRING_OFF = ((BIAS_OFF + MAX_CONVS * CH * 4 + 1023) / 1024) * 1024;
constexpr int WTILE_BYTES = CH * 64 * 2;      // [128 out][64 in] fp16 = 16 KiB
constexpr int WRING_BUFS = 3;
constexpr int LDS_BYTES = WRING_OFF + WRING_BUFS * WTILE_BYTES;
constexpr int KSTEPS_PER_CONV = 9 * (CH / 64);   // 18
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ inline half8 lds_read16(const lds_byte *base, int off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) half8 *>(base + off);
}

}  // namespace crl_tower
