// tower_gen.hpp -- ROUND-1 kernels for 64 / 256 filters and small batches (32x32x16 MFMA); since
// round 2 dispatched only by the tuning library (the product runs tower_x16.hpp, which reuses the
// helpers here).  The trunk structure of tower_pipe.hpp (LDS-resident boards, fp32
// residual stream in registers, 4-deep LDS-DMA weight ring, pinned hand-counted software pipeline,
// padded activation rows, conflict-free zero rows, head convs in the tail) templated on the number
// of filters F in {64, 128, 256}: every tower size of BASELINE.json (C2 6x64, C3/C4 10x128, C5
// 20x256) runs on a hand-written kernel.  F = 128 is normally served by the hand-tuned
// tower_pipe.hpp (3.5 % faster than this template at F = 128).
//
// Geometry per workgroup (512 threads, 8 waves, 2 per SIMD), Geo<F, NB> with NB boards resident:
//   F    NB  waves/board               wave tile        act. row  weight tile [F out][KT in]  sub-steps/tile
//   64    4  2 (position halves)       32 pos x 64 ch    272 B     64 x 64   ( 8 KiB)           4
//   128   4  2 (channel halves)        64 pos x 64 ch    272 B    128 x 64   (16 KiB)           4
//   256   2  4 (channel quarters)      64 pos x 64 ch    528 B    256 x 32   (16 KiB)           2
// and, for batches too small to give every CU a workgroup (<= 128 workgroups of the above), the
// same kernels with half the boards per workgroup and twice the workgroups:
//   64    2  4 (pos halves x ch halves) 32 pos x 32 ch
//   128   2  4 (pos halves x ch halves) 32 pos x 64 ch
//   256   1  8 (pos halves x ch quarters) 32 pos x 64 ch
// The stem reads the 128 input planes (channels 0..127 of the row; rows are at least 128 channels
// wide), every other layer F channels.  Biases are staged per layer (F floats).
#pragma once
#include "tower_pipe.hpp"

namespace crl_tower {

template <int F, int NB_>
struct Geo {
    static_assert(F == 64 || F == 128 || F == 256, "supported filter counts");
    static constexpr int NB = NB_;                      // boards per workgroup
    static constexpr int WPB = 8 / NB;                  // waves per board
    static constexpr int NT = (F == 64 && NB == 2) ? 1 : 2;     // 32-channel tiles per wave
    static constexpr int CG = F / (32 * NT);            // channel groups (waves splitting the channels)
    static constexpr int PH = WPB / CG;                 // position halves (waves splitting a board)
    static constexpr int MT = 2 / PH;                   // 32-position tiles per wave
    static_assert(NB * CG * PH == 8 && (PH == 1 || PH == 2), "8 waves per workgroup");
    static constexpr int KT = F == 256 ? 32 : 64;       // input channels per weight tile
    static constexpr int SPT = KT / 16;                 // 16-channel sub-steps per tile
    static constexpr int WROW = KT * 2;                 // bytes per weight-tile row
    static constexpr int WCH = WROW / 16;               // 16-B chunks per weight-tile row
    static constexpr int TILE_BYTES = F * WROW;         // 8 KiB (F = 64) or 16 KiB
    static constexpr int GL = TILE_BYTES / 8192;        // global_load_lds per thread per tile
    static constexpr int AROW = (F < 128 ? 128 : F) * 2 + 16;
    static constexpr int ABOARD = 64 * AROW;
    static constexpr int ZERO_OFF = NB * ABOARD;
    static constexpr int ZERO_BYTES = 16 * AROW;
    static constexpr int BIAS_OFF = ZERO_OFF + ZERO_BYTES;          // float [F], current layer
    static constexpr int WRING_OFF = ((BIAS_OFF + F * 4 + 1023) / 1024) * 1024;
    static constexpr int LDS_BYTES = WRING_OFF + PIPE_RING * TILE_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    // swizzle of a weight-tile row: XOR of the chunk index with row bits that differ inside a
    // ds_read_b128 lane group, so 16 lanes cover all 64 banks
    __device__ static int wswz(int row) { return WCH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};

template <class G>
__device__ inline void stage_wtile_gen(const unsigned char *wts, lds_byte *lds, int t, int tid)
{
    const unsigned char *src = wts + (size_t)t * G::TILE_BYTES;
    lds_byte *dst = lds + G::WRING_OFF + (t & (PIPE_RING - 1)) * G::TILE_BYTES;
    const int wave_base = tid & ~63;
#pragma unroll
    for (int j = 0; j < G::GL; j++) {
        const int idx = j * 512 + tid;                  // 16-B slot of the tile image
        const int row = idx / G::WCH, phys = idx % G::WCH;
        const int chunk = phys ^ G::wswz(row);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(src + row * G::WROW + chunk * 16),
            (__attribute__((address_space(3))) void *)(dst + (j * 512 + wave_base) * 16), 16, 0, 0);
    }
}

//   planes  fp16 [n_boards][64][128]
//   wts     fp16 tiles, consumption order [conv][tap][in-ch/KT][F out][KT in]
//   bias    f32 [n_convs][F];  head_w f32 [3][F];  head_b f32 [3]
//   out     f32 [n_boards][64][F] or nullptr;  head_out f32 [n_boards][192] or nullptr
template <int F, int NB, int BITS = 0>
__global__ __launch_bounds__(512, 2) void k_trunk_gen(const unsigned char *__restrict__ planes,
                                                       const unsigned char *__restrict__ wts,
                                                       const float *__restrict__ bias,
                                                       float *__restrict__ out, int n_blocks,
                                                       const float *__restrict__ head_w,
                                                       const float *__restrict__ head_b,
                                                       float *__restrict__ head_out)
{
    typedef Geo<F, NB> G;
    constexpr int MT = G::MT, NT = G::NT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    lds_byte *lds = (lds_byte *)lds_raw;
    const int lds_base = (int)(size_t)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int board = wave / G::WPB;
    const int obase = (32 * NT) * ((wave / G::PH) % G::CG);      // first output channel of this wave
    const int pbase = 32 * (wave % G::PH);              // first position of this wave
    const int r = lane & 31, h = lane >> 5;
    const int n_convs = 1 + 2 * n_blocks;
    const int tiles_stem = 9 * (128 / G::KT), tiles_conv = 9 * (F / G::KT);
    const int n_tiles = tiles_stem + 2 * n_blocks * tiles_conv;
    const size_t wg_board0 = (size_t)blockIdx.x * G::NB;

    stage_wtile_gen<G>(wts, lds, 0, tid);
    stage_wtile_gen<G>(wts, lds, 1, tid);
    stage_wtile_gen<G>(wts, lds, 2, tid);

    {   // planes (128 channels = 16 chunks per position) -> padded LDS rows; zero rows
        if constexpr (BITS) {
            expand_bitplanes<G::NB, G::AROW, G::ABOARD>(planes, lds, wg_board0, tid);
        } else {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(planes + wg_board0 * BOARD_BYTES);
#pragma unroll
            for (int i = 0; i < G::NB * 2; i++) {
                const int c16 = i * 512 + tid;
                const int p = (c16 >> 4) & 63, c = c16 & 15, b = c16 >> 10;
                u32x4 v = src[c16];
                *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(
                    lds + b * G::ABOARD + p * G::AROW + (c << 4)) = v;
            }
        }
        for (int i = tid; i < G::ZERO_BYTES / 16; i += 512)
            *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + G::ZERO_OFF + i * 16) =
                u32x4{0u, 0u, 0u, 0u};
    }
    wait_vmcnt<2 * G::GL>();                            // tile 0 landed (tiles 1,2 may be in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int px[MT], py[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) { const int p = pbase + 32 * mt + r; px[mt] = p & 7; py[mt] = p >> 3; }
    int waddr[NT][G::SPT];                              // weight fragment offset inside a tile
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int o = obase + 32 * nt + r;
#pragma unroll
        for (int s = 0; s < G::SPT; s++) waddr[nt][s] = o * G::WROW + (((2 * s + h) ^ G::wswz(o)) << 4);
    }

    f32x16 res[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) res[a][b][i] = 0.f;

    int t = 0;                                          // tile of the K-step being computed
    for (int conv = 0; conv < n_convs; conv++) {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int b = 0; b < NT; b++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
        // this layer's bias: fetched now, parked in LDS just before the epilogue barrier
        const float bias_reg = tid < F ? bias[conv * F + tid] : 0.f;

        auto tap_base = [&](int tap, int mt) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int yy = py[mt] + dy, xx = px[mt] + dx;
            const bool ok = ((unsigned)yy < 8u) && ((unsigned)xx < 8u);
            const int pp = yy * 8 + xx;
            return lds_base + (ok ? board * G::ABOARD + pp * G::AROW : G::ZERO_OFF + (pp & 15) * G::AROW) +
                   h * 16;
        };
        auto mfma_all = [&](const Frags &f) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w[nt], f.x[mt], acc[mt][nt], 0, 0, 0);
        };
        Frags f0, f1;
        int ab[2][MT];                                  // [current tap, next tap][mt]

        // one tap = NS sub-steps of 16 input channels (NS = Cin/16); sub-step i reads channel
        // block i of the activation rows and sub-step i % SPT of weight tile t_tap0 + i / SPT
        auto run_tap = [&](auto NSC, int tap, bool last_tap) {
            constexpr int NS = decltype(NSC)::value;
            const int t_tap0 = t;
            auto fetch = [&](auto IC, bool next_tap, Frags &f) {
                constexpr int i = decltype(IC)::value;
                const int tile = next_tap ? t_tap0 + NS / G::SPT : t_tap0 + i / G::SPT;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) f.x[mt] = lds_read16_asm<i * 32>(ab[next_tap ? 1 : 0][mt]);
                const int wb = lds_base + G::WRING_OFF + (tile & (PIPE_RING - 1)) * G::TILE_BYTES;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) f.w[nt] = lds_read16_asm<0>(wb + waddr[nt][i % G::SPT]);
            };
            if (tap == 0) fetch(std::integral_constant<int, 0>{}, false, f0);
            static_for<0, NS>([&](auto IC) {
                constexpr int i = decltype(IC)::value;
                constexpr int s = i % G::SPT;
                if constexpr (s == G::SPT - 1 - (G::SPT > 2 ? 1 : 0)) {
                    // publish tile t+1 before the sub-step that prefetches its first fragments
                    // (SPT = 4: before sub-step 2; SPT = 2: before sub-step 1); recycle tile t-1's slot
                    if (t + 2 < n_tiles) wait_vmcnt<G::GL>();
                    else wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 3 < n_tiles) stage_wtile_gen<G>(wts, lds, t + 3, tid);
                }
                constexpr bool wrap = i + 1 >= NS;
                bool issued = false;
                if (!wrap || !last_tap) {
                    issued = true;
                    if constexpr (wrap) {
                        if constexpr (i % 2 == 0) fetch(std::integral_constant<int, 0>{}, true, f1);
                        else fetch(std::integral_constant<int, 0>{}, true, f0);
                    } else {
                        if constexpr (i % 2 == 0) fetch(std::integral_constant<int, i + 1>{}, false, f1);
                        else fetch(std::integral_constant<int, i + 1>{}, false, f0);
                    }
                }
                // operands of THIS sub-step have landed; the MT+NT reads just issued may be in flight
                if (!issued) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else if constexpr (MT + NT == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if constexpr (MT + NT == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i % 2 == 0) mfma_all(f0);
                else mfma_all(f1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s == G::SPT - 1) t++;
            });
        };

#pragma unroll
        for (int mt = 0; mt < MT; mt++) ab[1][mt] = tap_base(0, mt);
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                ab[0][mt] = ab[1][mt];
                ab[1][mt] = tap_base(tap < 8 ? tap + 1 : 0, mt);
            }
            if (F == 128 || conv == 0) run_tap(std::integral_constant<int, 8>{}, tap, tap == 8);
            else run_tap(std::integral_constant<int, F / 16>{}, tap, tap == 8);
        }

        // ---- epilogue ------------------------------------------------------------------------------
        if (tid < F)
            *reinterpret_cast<__attribute__((address_space(3))) float *>(lds + G::BIAS_OFF + tid * 4) = bias_reg;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // all reads of the activation buffer done
        __builtin_amdgcn_sched_barrier(0);
        const bool is_stem = conv == 0;
        const bool is_conv2 = !is_stem && ((conv & 1) == 0);
        const bool keep_res = !is_stem && !is_conv2;
        const float relu_floor = is_stem ? -__builtin_inff() : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int p = pbase + 32 * mt + r;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int o0 = obase + 32 * nt + 8 * g + 4 * h;
                    const f32x4 bv = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(
                        lds + G::BIAS_OFF + o0 * 4);
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
                        lds + board * G::ABOARD + p * G::AROW + o0 * 2) = o16;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    if (out) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int p = pbase + 32 * mt + r;
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int o0 = obase + 32 * nt + 8 * g + 4 * h;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = res[mt][nt][4 * g + j];
                    *reinterpret_cast<f32x4 *>(out + ((wg_board0 + board) * 64 + p) * F + o0) = v;
                }
        }
    }

    if (head_out) {
        // a position's F channels live in CG waves x 2 lane halves: 2*CG partial sums per output
        constexpr int NC = 2 * G::CG;
        float part[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int k = 0; k < 3; k++) part[mt][k] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int o0 = obase + 32 * nt + 8 * g + 4 * h;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(head_w + k * F + o0);
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
#pragma unroll
                        for (int j = 0; j < 4; j++) part[mt][k] += res[mt][nt][4 * g + j] * wv[j];
                }
            }
        __attribute__((address_space(3))) float *scratch = (__attribute__((address_space(3))) float *)lds;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                scratch[(((board * 64 + pbase + 32 * mt + r) * 3) + k) * NC + (obase / (32 * NT)) * 2 + h] = part[mt][k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = tid; i < G::NB * 64 * 3; i += 512) {
            const int k = i % 3, bp = i / 3;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < NC; c++) v += scratch[i * NC + c];      // fixed order
            v += head_b[k];
            const size_t gb = wg_board0 + (bp >> 6);
            const int pos = bp & 63;
            head_out[gb * 192 + (k < 2 ? pos * 2 + k : 128 + pos)] = fmaxf(v, 0.f);
        }
    }
}

}  // namespace crl_tower
