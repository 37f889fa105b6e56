// tower_pipe.hpp -- PRODUCTION build of the fused residual trunk (128 filters) for gfx950.
//
// ROUND-1 PRODUCTION (32x32x16 MFMA), since round 2 compiled for dispatch only in the tuning library:
// the product runs tower_x16.hpp, which reuses this file's helpers (static_for, inline-asm LDS
// reads, plane expansion).  Design: tower_common.hpp (LDS-resident boards, fp32 residual stream in
// registers, LDS-DMA weight ring, transposed product, in-place epilogue).  tower.hpp is the first,
// straightforward build of it, the baseline of the ladder in profiles/r01/pmc_trunk_kernel.md;
// results are bit-identical to it.  Its rocprofv3 counters
// at the C3 shape showed the matrix pipe 51 % busy, waves parked in s_waitcnt/s_barrier 38 % of
// their cycles, ~4 VALU instructions per MFMA, the LDS array 39 % busy with 18 % conflicts.  What
// this build changes, in the order it paid:
//
//   1. ADDRESS-FREE inner loop.  The activation image is padded (272-byte rows: 16 consecutive
//      rows start on 16 different 4-bank groups, as the XOR swizzle did) so a fragment read is
//      `base(tap, position tile) + immediate`; weight fragment offsets are computed once per
//      kernel.  Non-MFMA VALU per MFMA: 3.9 -> 1.9.
//   2. PINNED SOFTWARE PIPELINE.  Fragments of sub-step u+1 are issued before the MFMAs of
//      sub-step u.  hipcc undoes that in two ways: its scheduler sinks the reads to their first
//      use, and its waitcnt pass answers a pinned prefetch with `lgkmcnt(0)` right behind the
//      newest reads.  So the order is pinned with sched_barrier(0) and the reads are inline-asm
//      `ds_read_b128` counted by hand (`s_waitcnt lgkmcnt(4)`: the 4 newest may be in flight).
//   3. MID-STEP BARRIER.  The weight ring is four tiles deep; the per-tile
//      `s_waitcnt vmcnt(2); s_barrier` sits before sub-step 2 of a K-step: it publishes tile t+1
//      two sub-steps before its first fragment read, so the matrix pipe does not drain at tile
//      boundaries (barrier cost now 1.7 %).
//   4. CONFLICT-FREE ZERO ROWS.  Off-board neighbours read one of 16 zero rows laid out like board
//      rows, i.e. on the bank group the lane would have used in-board (conflicts 18 % -> 5 %).
//
// Remaining costs (DIAG builds): weight staging 12.5 % (5 TB/s of L2->LDS traffic chip-wide; the
// only cure is more boards per workgroup, which LDS does not allow), epilogue + pipeline refill
// per layer ~3 %.  Without staging the kernel runs at the rate of a bare MFMA micro-benchmark
// with changing operands (1.5-1.6 PFLOP/s on this chip).
#pragma once
#include "r1_common.hpp"

namespace crl_tower {

struct Frags { half8 x[2]; half8 w[2]; };          // operands of one 16-channel sub-step

constexpr int P2_AROW = ROW_BYTES + 16;                             // padded activation row
constexpr int P2_ABOARD = 64 * P2_AROW;
constexpr int P2_ZERO_OFF = BOARDS_PER_WG * P2_ABOARD;              // 16 zero rows
constexpr int P2_ZERO_BYTES = 16 * P2_AROW;
constexpr int P2_BIAS_OFF = P2_ZERO_OFF + P2_ZERO_BYTES;            // float [MAX_CONVS][128]
constexpr int P2_WRING_OFF = ((P2_BIAS_OFF + MAX_CONVS * CH * 4 + 1023) / 1024) * 1024;
constexpr int P2_LDS_BYTES = P2_WRING_OFF + PIPE_RING * WTILE_BYTES;
static_assert(P2_LDS_BYTES <= 160 * 1024, "LDS budget");

// stage weight tile t into ring slot t & 3: LDS image linear per wave instruction, XOR swizzle
// (chunk ^ ((row >> 1) & 7)) applied to the per-lane SOURCE address
__device__ inline void stage_wtile_p2(const unsigned char *wts, lds_byte *lds, int t, int tid)
{
    const unsigned char *src = wts + (size_t)t * WTILE_BYTES;
    lds_byte *dst = lds + P2_WRING_OFF + (t & (PIPE_RING - 1)) * WTILE_BYTES;
    const int wave_base = tid & ~63;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int idx = j * 512 + tid;
        const int row = idx >> 3, phys = idx & 7;
        const int chunk = phys ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(src + row * 128 + chunk * 16),
            (__attribute__((address_space(3))) void *)(dst + (j * 512 + wave_base) * 16), 16, 0, 0);
    }
}

// DIAG = 0: production.  Timing-only builds (WRONG results): bit 0 = no weight staging in the loop,
// bit 1 = no per-tile barrier, bit 4 (16) = activation fragments read for the dx = 0 taps only (what a
// cross-lane generation of the dx = +-1 fragments would leave).  Bit 2 (correct results) = staggered staging: waves 0-3 stage at the
// mid-step barrier, waves 4-7 half a K-step later (measured 2 % slower than staging together).
// BITS = 1: `planes` holds 128 plane bitboards per board (expand_bitplanes) instead of fp16 planes.
template <int DIAG = 0, int BITS = 0>
__global__ __launch_bounds__(512, 2) void k_trunk128_pipe(const unsigned char *__restrict__ planes,
                                                           const unsigned char *__restrict__ wts,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ out, int n_blocks,
                                                           const float *__restrict__ head_w,
                                                           const float *__restrict__ head_b,
                                                           float *__restrict__ head_out)
{
#ifndef CRL_TUNING
    static_assert(DIAG == 0, "timing-only variants exist in the tuning library only");
#endif
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    lds_byte *lds = (lds_byte *)lds_raw;
    const int lds_base = (int)(size_t)lds;              // LDS byte address of the dynamic array
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int board = wave >> 1, nh = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int n_convs = 1 + 2 * n_blocks;
    const int n_tiles = n_convs * KSTEPS_PER_CONV;
    const size_t wg_board0 = (size_t)blockIdx.x * BOARDS_PER_WG;

    // ---- weight stream prologue: three tiles in flight ---------------------------------------------
    stage_wtile_p2(wts, lds, 0, tid);
    stage_wtile_p2(wts, lds, 1, tid);
    stage_wtile_p2(wts, lds, 2, tid);

    // ---- planes -> padded LDS image; zero rows; biases -------------------------------------------------
    {
        if constexpr (BITS) {
            expand_bitplanes<BOARDS_PER_WG, P2_AROW, P2_ABOARD>(planes, lds, wg_board0, tid);
        } else {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(planes + wg_board0 * BOARD_BYTES);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int c16 = i * 512 + tid;          // 16-B chunk of the 64-KiB input block
                const int p = (c16 >> 4) & 63, c = c16 & 15, b = c16 >> 10;
                u32x4 v = src[c16];
                *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(
                    lds + b * P2_ABOARD + p * P2_AROW + (c << 4)) = v;
            }
        }
        if (tid < P2_ZERO_BYTES / 16)
            *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + P2_ZERO_OFF + tid * 16) =
                u32x4{0u, 0u, 0u, 0u};
        for (int i = tid; i < n_convs * CH; i += 512)
            *reinterpret_cast<__attribute__((address_space(3))) float *>(lds + P2_BIAS_OFF + i * 4) = bias[i];
    }
    // tile 0 landed (tiles 1,2 may be in flight), planes/bias/zero rows written: publish
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int px[2], py[2];                                   // the lane's two positions p = 32*mt + r
#pragma unroll
    for (int mt = 0; mt < 2; mt++) { const int p = 32 * mt + r; px[mt] = p & 7; py[mt] = p >> 3; }
    int waddr[2][4];                                    // weight fragment offset inside a tile
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int o = 64 * nh + 32 * nt + r;
#pragma unroll
        for (int s = 0; s < 4; s++) waddr[nt][s] = o * 128 + (((2 * s + h) ^ ((o >> 1) & 7)) << 4);
    }

    f32x16 res[2][2];                                   // fp32 residual stream [mt][nt]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) res[a][b][i] = 0.f;

    int t = 0;                                          // tile of the K-step being computed
    for (int conv = 0; conv < n_convs; conv++) {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;

        const int t_conv0 = t;
        // per-tap activation base (one VGPR per position tile): everything else is an immediate
        auto tap_base = [&](int tap, int mt) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int yy = py[mt] + dy, xx = px[mt] + dx;
            const bool ok = ((unsigned)yy < 8u) && ((unsigned)xx < 8u);
            const int pp = yy * 8 + xx;
            return lds_base + (ok ? board * P2_ABOARD + pp * P2_AROW : P2_ZERO_OFF + (pp & 15) * P2_AROW) +
                   h * 16;
        };
        // sub-step i (0..23) of tap block blk (3 taps = 24 sub-steps): tap 3*blk + i/8, channel half
        // (i/4)&1, 16-channel step i&3; weight tile t_conv0 + 6*blk + i/4
        int ab[4][2];                                   // taps of the current block + first of the next
        auto fetch = [&](auto IC, int blk, bool next_blk, Frags &f) {
            constexpr int i = decltype(IC)::value;
            constexpr int kc = (i >> 2) & 1, s4 = i & 3;
            if constexpr (!(DIAG & 16) || (i >> 3) == 1) {          // DIAG 16: only the dx = 0 taps read x
                f.x[0] = lds_read16_asm<(kc * 8 + 2 * s4) * 16>(ab[next_blk ? 3 : (i >> 3)][0]);
                f.x[1] = lds_read16_asm<(kc * 8 + 2 * s4) * 16>(ab[next_blk ? 3 : (i >> 3)][1]);
            }
            const int wb = lds_base + P2_WRING_OFF +
                           ((t_conv0 + 6 * blk + (i >> 2)) & (PIPE_RING - 1)) * WTILE_BYTES;
            f.w[0] = lds_read16_asm<0>(wb + waddr[0][s4]);
            f.w[1] = lds_read16_asm<0>(wb + waddr[1][s4]);
        };
        auto mfma4 = [&](const Frags &f) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w[nt], f.x[mt], acc[mt][nt], 0, 0, 0);
        };

        Frags f0, f1;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) ab[3][mt] = tap_base(0, mt);
        for (int blk = 0; blk < 3; blk++) {             // 3 blocks x 24 sub-steps (3 taps each)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                ab[0][mt] = ab[3][mt];
                ab[1][mt] = tap_base(3 * blk + 1, mt);
                ab[2][mt] = tap_base(3 * blk + 2, mt);
                ab[3][mt] = tap_base(blk < 2 ? 3 * blk + 3 : 0, mt);
            }
            if (blk == 0) fetch(std::integral_constant<int, 0>{}, 0, false, f0);
            static_for<0, 24>([&](auto IC) {
                constexpr int i = decltype(IC)::value;
                if constexpr ((i & 3) == 2) {
                    // middle of K-step t: publish tile t+1 (its first read is issued one sub-step
                    // before its K-step starts), recycle the buffer of tile t-1 for tile t+3
                    if (!(DIAG & 32)) {                  // DIAG 32 (timing only): do not wait for the DMA
                        if (t + 2 < n_tiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    if (!(DIAG & 2)) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DIAG & 1) && t + 3 < n_tiles && (!(DIAG & 4) || wave < 4))
                        stage_wtile_p2(wts, lds, t + 3, tid);
                }
                if constexpr ((i & 3) == 0) {
                    if (!(DIAG & 1) && (DIAG & 4) && wave >= 4 && t > 0 && t + 2 < n_tiles)
                        stage_wtile_p2(wts, lds, t + 2, tid);
                }
                // prefetch the next sub-step into the register group the previous one released
                constexpr bool wrap = i + 1 >= 24;
                bool issued = false;
                if (!wrap || blk < 2) {
                    issued = true;
                    if constexpr (wrap) {
                        if constexpr (i % 2 == 0) fetch(std::integral_constant<int, 0>{}, blk + 1, true, f1);
                        else fetch(std::integral_constant<int, 0>{}, blk + 1, true, f0);
                    } else {
                        if constexpr (i % 2 == 0) fetch(std::integral_constant<int, i + 1>{}, blk, false, f1);
                        else fetch(std::integral_constant<int, i + 1>{}, blk, false, f0);
                    }
                }
                // operands of THIS sub-step have landed; the 4 reads just issued may be in flight
                constexpr int nxt = wrap ? 0 : i + 1;
                if (!issued) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else if constexpr ((DIAG & 16) && (nxt >> 3) != 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i % 2 == 0) mfma4(f0);
                else mfma4(f1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((i & 3) == 3) t++;
            });
        }

        // ---- epilogue: every wave has finished reading the activation buffer ----------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool is_stem = conv == 0;
        const bool is_conv2 = !is_stem && ((conv & 1) == 0);     // convs 1,3,.. = conv1; 2,4,.. = conv2
        const bool keep_res = !is_stem && !is_conv2;             // conv1 leaves the skip stream alone
        const float relu_floor = is_stem ? -__builtin_inff() : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int p = 32 * mt + r;
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int o0 = 64 * nh + 32 * nt + 8 * g + 4 * h;     // 4 consecutive channels
                    const f32x4 bv = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(
                        lds + P2_BIAS_OFF + (conv * CH + o0) * 4);
                    half4 o16;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        // wave-uniform selects instead of branches: stem = bias only (no BN, no
                        // activation, model.py:33-34); conv1 = ReLU; conv2 = +skip, ReLU
                        const float skip = is_conv2 ? res[mt][nt][4 * g + j] : 0.f;
                        float v = (acc[mt][nt][4 * g + j] + bv[j]) + skip;
                        v = fmaxf(v, relu_floor);
                        res[mt][nt][4 * g + j] = keep_res ? res[mt][nt][4 * g + j] : v;
                        o16[j] = (_Float16)v;
                    }
                    *reinterpret_cast<__attribute__((address_space(3))) half4 *>(
                        lds + board * P2_ABOARD + p * P2_AROW + o0 * 2) = o16;
                }
            }
        }
        // the rewritten activations must be visible before the next layer's first fragment reads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- optional trunk output: fp32 residual stream -> global [board][pos][ch] -----------------------
    if (out) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int p = 32 * mt + r;
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int o0 = 64 * nh + 32 * nt + 8 * g + 4 * h;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = res[mt][nt][4 * g + j];
                    *reinterpret_cast<f32x4 *>(out + ((wg_board0 + board) * 64 + p) * CH + o0) = v;
                }
        }
    }

    // ---- head 1x1 convolutions (policy 2 ch, value 1 ch; BN folded) + ReLU, in fp32 -----------------
    // model.py:40-42,51-55.  A position's 128 channels are spread over 2 waves x 2 lane halves: each
    // lane reduces its 32 channels, the 4 partial sums meet in LDS (the activation buffer is dead
    // now) and are added in a FIXED order (no float atomics: results are reproducible).
    if (head_out) {
        float part[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int k = 0; k < 3; k++) part[mt][k] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int o0 = 64 * nh + 32 * nt + 8 * g + 4 * h;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(head_w + k * CH + o0);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        part[0][k] += res[0][nt][4 * g + j] * wv[j];
                        part[1][k] += res[1][nt][4 * g + j] * wv[j];
                    }
                }
            }
        __attribute__((address_space(3))) float *scratch = (__attribute__((address_space(3))) float *)lds;
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                scratch[(((board * 64 + 32 * mt + r) * 3) + k) * 4 + nh * 2 + h] = part[mt][k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = tid; i < BOARDS_PER_WG * 64 * 3; i += 512) {
            const int k = i % 3, bp = i / 3;           // bp = board * 64 + position
            const f32x4 c = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(scratch + i * 4);
            const float v = (((c[0] + c[1]) + c[2]) + c[3]) + head_b[k];
            // per board 192 floats: [0,128) = policy head in Keras Flatten order (pos*2 + ch),
            // [128,192) = value head (pos): both dense layers read them without a gather
            const size_t gb = wg_board0 + (bp >> 6);
            const int pos = bp & 63;
            head_out[gb * 192 + (k < 2 ? pos * 2 + k : 128 + pos)] = fmaxf(v, 0.f);
        }
    }
}

}  // namespace crl_tower
