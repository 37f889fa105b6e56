// tower_pipe.hpp -- PRODUCTION build of the fused trunk: tower.hpp's structure, software-
// pipelined, with an ADDRESS-FREE inner loop: the activation image is
// padded (272-byte rows) instead of XOR-swizzled, so every activation fragment read is one
// per-tap base register + an immediate offset, and the weight fragment offsets are computed once
// per kernel.  tower.hpp spends ~4 VALU instructions per MFMA on fragment addresses; with two
// waves per SIMD sharing one issue port that is what keeps the matrix pipe at ~50 %.
// (same math and weight tile format as tower.hpp; results are bit-identical to it).
//
// What tower.hpp's rocprofv3 counters showed at the C3 shape (profiles/r01): MFMA pipe 51 % busy,
// waves parked in s_waitcnt/s_barrier 38 % of their cycles, LDS array only 39 % busy -- the loss
// is LDS *latency* at the start of every 16-channel sub-step and of every K-step, not bandwidth.
// This build hides it:
//   * operand fragments are fetched TWO sub-steps ahead into a rotating set of three register
//     groups (compile-time rotation: the 24 sub-steps of three taps are fully unrolled);
//   * the weight ring is four tiles deep and the per-tile `s_waitcnt vmcnt; s_barrier` sits in
//     the MIDDLE of a K-step (before sub-step 2): it publishes tile t+1 two sub-steps before its
//     first fragment read is issued, so MFMAs of tile t are still queued behind the barrier and
//     the matrix pipe does not drain at K-step boundaries;
//   * the only full drains left are the two barriers around each layer's epilogue.
#pragma once
#include <type_traits>
#include "tower.hpp"

namespace crl_tower {

constexpr int PIPE_RING = 4;                       // weight tiles in the LDS ring
struct Frags { half8 x[2]; half8 w[2]; };          // operands of one 16-channel sub-step

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ds_read_b128 the compiler does not see: hipcc's waitcnt pass answers a pinned prefetch with
// `s_waitcnt lgkmcnt(0)` right behind the newest reads, i.e. it waits for the data of the NEXT
// sub-step before issuing the MFMAs of the current one.  These reads are counted by hand
// (`s_waitcnt lgkmcnt(4)` = the 4 newest may still be in flight).
template <int OFF>
__device__ __forceinline__ half8 lds_read16_asm(int addr)
{
    half8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// padded activation image: row stride 272 B = 68 dwords -> 16 rows with distinct (row mod 16)
// start on 16 different 4-bank groups, exactly what the XOR swizzle achieved
constexpr int P2_AROW = ROW_BYTES + 16;
constexpr int P2_ABOARD = 64 * P2_AROW;
// off-board neighbours read zeros.  A single zero row would sit on ONE bank group and collide with
// whichever in-board lane of the same ds_read_b128 group owns it (rocprofv3: 18 % of the LDS cycles
// were conflicts); 16 zero rows laid out like board rows let an off-board lane read the row whose
// bank group it would have used in-board: conflict-free by construction.
constexpr int P2_ZERO_OFF = BOARDS_PER_WG * P2_ABOARD;
constexpr int P2_ZERO_BYTES = 16 * P2_AROW;                         // 4352 B
constexpr int P2_BIAS_OFF = P2_ZERO_OFF + P2_ZERO_BYTES;
constexpr int P2_WRING_OFF = ((P2_BIAS_OFF + MAX_CONVS * CH * 4 + 1023) / 1024) * 1024;
constexpr int P2_LDS_BYTES = P2_WRING_OFF + PIPE_RING * WTILE_BYTES;
static_assert(P2_LDS_BYTES <= 160 * 1024, "LDS budget");

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



// DIAG (timing only, WRONG results): 1 = no weight staging in the loop, 2 = no per-tile barrier,
// 3 = neither
template <int DIST, int ASMRD, int DIAG = 0>
__global__ __launch_bounds__(512, 2) void k_trunk128_pipe(const unsigned char *__restrict__ planes,
                                                           const unsigned char *__restrict__ wts,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ out, int n_blocks,
                                                           const float *__restrict__ head_w,
                                                           const float *__restrict__ head_b,
                                                           float *__restrict__ head_out)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    lds_byte *lds = (lds_byte *)lds_raw;
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

    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(planes + wg_board0 * BOARD_BYTES);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c16 = i * 512 + tid;
            const int p = (c16 >> 4) & 63, c = c16 & 15, b = c16 >> 10;
            u32x4 v = src[c16];
            *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(
                lds + b * P2_ABOARD + p * P2_AROW + (c << 4)) = v;
        }
        if (tid < P2_ZERO_BYTES / 16)
            *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + P2_ZERO_OFF + tid * 16) =
                u32x4{0u, 0u, 0u, 0u};
        for (int i = tid; i < n_convs * CH; i += 512)
            *reinterpret_cast<__attribute__((address_space(3))) float *>(lds + P2_BIAS_OFF + i * 4) = bias[i];
    }
    // tile 0 landed (tiles 1,2 may be in flight), planes/bias/zero row written: publish
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int px[2], py[2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) { const int p = 32 * mt + r; px[mt] = p & 7; py[mt] = p >> 3; }
    int waddr[2][4];                                    // weight fragment offset inside a tile
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int o = 64 * nh + 32 * nt + r;
#pragma unroll
        for (int s = 0; s < 4; s++) waddr[nt][s] = o * 128 + (((2 * s + h) ^ ((o >> 1) & 7)) << 4);
    }

    f32x16 res[2][2];
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

        // fragment fetch of sub-step i (0..23) of tap block blk (3 taps = 24 sub-steps): tap
        // 3*blk + i/8, channel half (i/4)&1, 16-channel step i&3; weight tile t_conv0 + 6*blk + i/4
        const int t_conv0 = t;
        // per-tap activation base (one VGPR per position tile): everything else is an immediate
        auto tap_base = [&](int tap, int mt) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int yy = py[mt] + dy, xx = px[mt] + dx;
            const bool ok = ((unsigned)yy < 8u) && ((unsigned)xx < 8u);
            const int pp = yy * 8 + xx;
            return (ok ? board * P2_ABOARD + pp * P2_AROW : P2_ZERO_OFF + (pp & 15) * P2_AROW) + h * 16;
        };
        int ab[4][2];                                   // taps of the current block + first of the next
        auto fetch = [&](int blk, int i, bool next_blk, Frags &f) {
            const int kc = (i >> 2) & 1, s = i & 3;
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
                f.x[mt] = lds_read16(lds, ab[next_blk ? 3 : (i >> 3)][mt] + (kc * 8 + 2 * s) * 16);
            const lds_byte *wbuf =
                lds + P2_WRING_OFF + ((t_conv0 + 6 * blk + (i >> 2)) & (PIPE_RING - 1)) * WTILE_BYTES;
#pragma unroll
            for (int nt = 0; nt < 2; nt++) f.w[nt] = lds_read16(wbuf, waddr[nt][s]);
        };
        auto mfma4 = [&](const Frags &f) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w[nt], f.x[mt], acc[mt][nt], 0, 0, 0);
        };

        Frags f0, f1, f2;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) ab[3][mt] = tap_base(0, mt);
        if constexpr (ASMRD) {
            // hand-counted variant (prefetch distance 1)
            const int lds_base = (int)(size_t)lds;      // LDS byte address of the dynamic array
            auto fetch_asm = [&](auto IC, int blk, bool next_blk, Frags &f) {
                constexpr int i = decltype(IC)::value;
                constexpr int kc = (i >> 2) & 1, s4 = i & 3;
                f.x[0] = lds_read16_asm<(kc * 8 + 2 * s4) * 16>(lds_base + ab[next_blk ? 3 : (i >> 3)][0]);
                f.x[1] = lds_read16_asm<(kc * 8 + 2 * s4) * 16>(lds_base + ab[next_blk ? 3 : (i >> 3)][1]);
                const int wb = lds_base + P2_WRING_OFF +
                               ((t_conv0 + 6 * blk + (i >> 2)) & (PIPE_RING - 1)) * WTILE_BYTES;
                f.w[0] = lds_read16_asm<0>(wb + waddr[0][s4]);
                f.w[1] = lds_read16_asm<0>(wb + waddr[1][s4]);
            };
            for (int blk = 0; blk < 3; blk++) {
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    ab[0][mt] = ab[3][mt];
                    ab[1][mt] = tap_base(3 * blk + 1, mt);
                    ab[2][mt] = tap_base(3 * blk + 2, mt);
                    ab[3][mt] = tap_base(blk < 2 ? 3 * blk + 3 : 0, mt);
                }
                if (blk == 0) fetch_asm(std::integral_constant<int, 0>{}, 0, false, f0);
                static_for<0, 24>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    if constexpr ((i & 3) == 2) {
                        if (t + 2 < n_tiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (!(DIAG & 2)) __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        // DIAG & 4 (correct results): stagger the staging -- waves 0-3 here, waves 4-7
                        // half a K-step later -- so that the two waves of a SIMD do not issue their
                        // LDS-DMA together.  Measured 2 % SLOWER than staging together: kept off.
                        if (!(DIAG & 1) && t + 3 < n_tiles && (!(DIAG & 4) || wave < 4))
                            stage_wtile_p2(wts, lds, t + 3, tid);
                    }
                    if constexpr ((i & 3) == 0) {
                        // start of K-step t: tile t+2's buffer was recycled by the last mid-step barrier
                        if (!(DIAG & 1) && (DIAG & 4) && wave >= 4 && t > 0 && t + 2 < n_tiles)
                            stage_wtile_p2(wts, lds, t + 2, tid);
                    }
                    constexpr bool wrap = i + 1 >= 24;
                    bool issued = false;
                    if (!wrap || blk < 2) {
                        issued = true;
                        if constexpr (wrap) {
                            if constexpr (i % 2 == 0) fetch_asm(std::integral_constant<int, 0>{}, blk + 1, true, f1);
                            else fetch_asm(std::integral_constant<int, 0>{}, blk + 1, true, f0);
                        } else {
                            if constexpr (i % 2 == 0) fetch_asm(std::integral_constant<int, i + 1>{}, blk, false, f1);
                            else fetch_asm(std::integral_constant<int, i + 1>{}, blk, false, f0);
                        }
                    }
                    // operands of THIS sub-step have landed; the 4 reads just issued may be in flight
                    if (issued) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (i % 2 == 0) mfma4(f0);
                    else mfma4(f1);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr ((i & 3) == 3) t++;
                });
            }
        } else {
        for (int blk = 0; blk < 3; blk++) {             // 3 blocks x 24 sub-steps (3 taps each)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                ab[0][mt] = ab[3][mt];
                ab[1][mt] = tap_base(3 * blk + 1, mt);
                ab[2][mt] = tap_base(3 * blk + 2, mt);
                ab[3][mt] = tap_base(blk < 2 ? 3 * blk + 3 : 0, mt);
            }
            if (blk == 0) {
                fetch(0, 0, false, f0);
                if (DIST == 2) fetch(0, 1, false, f1);
            }
#pragma unroll
            for (int i = 0; i < 24; i++) {
                if ((i & 3) == 2) {
                    // middle of K-step t: publish tile t+1 (its first read is issued DIST sub-steps
                    // before its K-step starts), recycle the buffer of tile t-1 for tile t+3
                    if (t + 2 < n_tiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 3 < n_tiles) stage_wtile_p2(wts, lds, t + 3, tid);
                }
                // prefetch sub-step +DIST into the register group that sub-step -1 just released
                const bool wrap = i + DIST >= 24;
                if (!wrap || blk < 2) {
                    const int pb = wrap ? blk + 1 : blk, pi = wrap ? i + DIST - 24 : i + DIST;
                    if (DIST == 2) {
                        if (i % 3 == 0) fetch(pb, pi, wrap, f2);
                        else if (i % 3 == 1) fetch(pb, pi, wrap, f0);
                        else fetch(pb, pi, wrap, f1);
                    } else {
                        if (i % 2 == 0) fetch(pb, pi, wrap, f1);
                        else fetch(pb, pi, wrap, f0);
                    }
                }
                // pin the software pipeline: hipcc otherwise sinks the prefetch down to its first
                // use (lgkmcnt(0) in front of every MFMA group) to save registers
                __builtin_amdgcn_sched_barrier(0);
                if (DIST == 2) {
                    if (i % 3 == 0) mfma4(f0);
                    else if (i % 3 == 1) mfma4(f1);
                    else mfma4(f2);
                } else {
                    if (i % 2 == 0) mfma4(f0);
                    else mfma4(f1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if ((i & 3) == 3) t++;
            }
        }

        }

        // ---- epilogue: every wave has finished reading the activation buffer ----------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool is_stem = conv == 0;
        const bool is_conv2 = !is_stem && ((conv & 1) == 0);
        const bool keep_res = !is_stem && !is_conv2;
        const float relu_floor = is_stem ? -__builtin_inff() : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int p = 32 * mt + r;
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int o0 = 64 * nh + 32 * nt + 8 * g + 4 * h;
                    const f32x4 bv = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(
                        lds + P2_BIAS_OFF + (conv * CH + o0) * 4);
                    half4 o16;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
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
            const int k = i % 3, bp = i / 3;
            const f32x4 c = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(scratch + i * 4);
            const float v = (((c[0] + c[1]) + c[2]) + c[3]) + head_b[k];
            const size_t gb = wg_board0 + (bp >> 6);
            const int pos = bp & 63;
            head_out[gb * 192 + (k < 2 ? pos * 2 + k : 128 + pos)] = fmaxf(v, 0.f);
        }
    }
}

}  // namespace crl_tower
