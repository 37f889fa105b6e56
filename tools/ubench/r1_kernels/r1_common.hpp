// r1_common.hpp -- LDS layout constants of the round-1 trunk kernels (32x32x16 MFMA), which are no
// longer part of the product: they are compiled only by tools/ubench/trunk_variants.hip as the
// baselines of the tuning ladder (profiles/r01/pmc_trunk_kernel.md).  Shared types and helpers
// come from chessrl_amd/csrc/tower_common.hpp.
#pragma once
#include "tower_common.hpp"

namespace crl_tower {

constexpr int ACT_BYTES = BOARDS_PER_WG * BOARD_BYTES;   // 64 KiB
constexpr int ZERO_OFF = ACT_BYTES;           // 256 B of zeros
constexpr int BIAS_OFF = ZERO_OFF + 256;      // float [MAX_CONVS][128]
constexpr int WRING_OFF = ((BIAS_OFF + MAX_CONVS * CH * 4 + 1023) / 1024) * 1024;
constexpr int WTILE_BYTES = CH * 64 * 2;      // [128 out][64 in] fp16 = 16 KiB
constexpr int WRING_BUFS = 3;
constexpr int LDS_BYTES = WRING_OFF + WRING_BUFS * WTILE_BYTES;
constexpr int KSTEPS_PER_CONV = 9 * (CH / 64);   // 18
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

}  // namespace crl_tower
