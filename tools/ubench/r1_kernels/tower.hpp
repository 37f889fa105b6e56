// tower.hpp -- FIRST BUILD of the fused 128-filter trunk (design: tower_common.hpp) and its
// timing-only diagnostic variants: the baseline of the ladder in profiles/r01/pmc_trunk_kernel.md.
// TUNING LIBRARY ONLY: compiled under -DCRL_TUNING (tools/trunk_bench.py), never into the product
// libchessrl_hip.so -- several variants here produce WRONG results by construction.
#pragma once
#ifndef CRL_TUNING
#error "tower.hpp belongs to the tuning build (-DCRL_TUNING) only"
#endif
#include "r1_common.hpp"

namespace crl_tower {

// stage weight tile `t` into ring buffer t % 3: 2 x 16 B per thread, LDS image linear per wave
// instruction, XOR swizzle applied to the per-lane SOURCE address
__device__ inline void stage_wtile(const unsigned char *wts, lds_byte *lds, int t, int tid)
{
    const unsigned char *src = wts + (size_t)t * WTILE_BYTES;
    lds_byte *dst = lds + WRING_OFF + (t % WRING_BUFS) * WTILE_BYTES;
    const int wave_base = tid & ~63;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int idx = j * 512 + tid;                 // 16-B slot in the tile image
        const int row = idx >> 3, phys = idx & 7;
        const int chunk = phys ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(src + row * 128 + chunk * 16),
            (__attribute__((address_space(3))) void *)(dst + (j * 512 + wave_base) * 16), 16, 0, 0);
    }
}

// One workgroup = 4 boards through stem + n_blocks residual blocks.
//   planes : fp16 [n_boards][64][128] (NHWC, channel 127 zero pad)
//   wts    : fp16 tiles, consumption order [conv][tap][kc][128 out][64 in]
//   bias   : f32 [n_convs][128]
//   out    : f32 [n_boards][64][128]  trunk output (after the last block's ReLU), or nullptr
//   head_w : f32 [3][128], head_b f32 [3]: folded 1x1 head convs (policy ch 0,1; value ch 2)
//   head_out: f32 [n_boards][192] = ReLU(head convs): 128 policy (pos*2+ch) + 64 value, or nullptr
// VAR: 0 = production.  100.. = timing-only diagnostics (WRONG results): 100 no K-step barrier,
// 101 no weight staging in the loop, 102 weight fragments read once, 103 activation fragments
// read once.  1.. = tuning variants (correct results).
template <int VAR>
__global__ __launch_bounds__(512, 2) void k_trunk128(const unsigned char *__restrict__ planes,
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

    // ---- weight stream prologue: two tiles in flight -------------------------------------------
    stage_wtile(wts, lds, 0, tid);
    stage_wtile(wts, lds, 1, tid);

    // ---- planes -> swizzled LDS image; zero row; biases ------------------------------------------
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(planes + wg_board0 * BOARD_BYTES);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int q = i * 512 + tid;               // 16-B chunk of the 64-KiB block
            const int p = (q >> 4) & 63, c = q & 15, b = q >> 10;
            u32x4 v = src[q];
            *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(
                lds + b * BOARD_BYTES + p * ROW_BYTES + ((c ^ (p & 15)) << 4)) = v;
        }
        if (tid < 16)
            *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + ZERO_OFF + tid * 16) =
                u32x4{0u, 0u, 0u, 0u};
        for (int i = tid; i < n_convs * CH; i += 512)
            *reinterpret_cast<__attribute__((address_space(3))) float *>(lds + BIAS_OFF + i * 4) = bias[i];
    }

    // per-lane geometry of the two position tiles (mt = 0,1): p = 32*mt + r
    int px[2], py[2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) { const int p = 32 * mt + r; px[mt] = p & 7; py[mt] = p >> 3; }
    // weight fragment rows (out-channels) of the two channel tiles (nt = 0,1)
    int wrow_off[2], wrow_swz[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int o = 64 * nh + 32 * nt + r;
        wrow_off[nt] = o * 128;
        wrow_swz[nt] = (o >> 1) & 7;
    }

    f32x16 res[2][2];                                   // fp32 residual stream, [mt][nt]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) res[a][b][i] = 0.f;

    int t = 0;                                          // weight tile counter (consumption order)
    for (int conv = 0; conv < n_convs; conv++) {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;

        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            int abase[2], aswz[2];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                const int yy = py[mt] + dy, xx = px[mt] + dx;
                const bool ok = ((unsigned)yy < 8u) && ((unsigned)xx < 8u);
                const int pp = yy * 8 + xx;
                abase[mt] = ok ? board * BOARD_BYTES + pp * ROW_BYTES : ZERO_OFF;
                aswz[mt] = ok ? (pp & 15) : 0;
            }
#pragma unroll
            for (int kc = 0; kc < 2; kc++) {
                // ---- one K-step: tile t is in ring buffer t % 3 -----------------------------------
                // my part of tile t has landed (tile t+1 may still be in flight); my LDS writes
                // (epilogue of the previous layer) have completed
                half8 xfa[4][2], wfa[4][2];
                if (VAR == 3) {
                    // the activation buffer is stable for the whole layer: its fragments need not
                    // wait for the weight tile's barrier
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int mt = 0; mt < 2; mt++)
                            xfa[s][mt] = lds_read16(lds, abase[mt] + (((kc * 8 + 2 * s + h) ^ aswz[mt]) << 4));
                    if (t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    if (t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                if (VAR != 100 && VAR != 104 && VAR != 105 && VAR != 106) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (VAR != 101 && VAR != 104 && VAR != 105 && VAR != 106 && t + 2 < n_tiles) stage_wtile(wts, lds, t + 2, tid);
                const lds_byte *wbuf = lds + WRING_OFF + (t % WRING_BUFS) * WTILE_BYTES;
                if (VAR == 2 || VAR == 3) {
                    if (VAR == 2) {
#pragma unroll
                        for (int s = 0; s < 4; s++)
#pragma unroll
                            for (int mt = 0; mt < 2; mt++)
                                xfa[s][mt] = lds_read16(lds, abase[mt] + (((kc * 8 + 2 * s + h) ^ aswz[mt]) << 4));
                    }
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int nt = 0; nt < 2; nt++)
                            wfa[s][nt] = lds_read16(wbuf, wrow_off[nt] + (((2 * s + h) ^ wrow_swz[nt]) << 4));
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int mt = 0; mt < 2; mt++)
#pragma unroll
                            for (int nt = 0; nt < 2; nt++)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfa[s][nt], xfa[s][mt], acc[mt][nt], 0, 0, 0);
                    t++;
                    continue;
                }
                half8 xf[2], wf[2];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if ((VAR != 103 && VAR != 104 && VAR != 105) || (s == 0 && (VAR != 105 || t == 0))) {
#pragma unroll
                        for (int mt = 0; mt < 2; mt++)
                            xf[mt] = lds_read16(lds, abase[mt] + (((kc * 8 + 2 * s + h) ^ aswz[mt]) << 4));
                    }
                    if ((VAR != 102 && VAR != 104 && VAR != 105) || (s == 0 && (VAR != 105 || t == 0))) {
#pragma unroll
                        for (int nt = 0; nt < 2; nt++)
                            wf[nt] = lds_read16(wbuf, wrow_off[nt] + (((2 * s + h) ^ wrow_swz[nt]) << 4));
                    }
                    if (VAR == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int nt = 0; nt < 2; nt++)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
                    if (VAR == 1) __builtin_amdgcn_s_setprio(0);
                }
                t++;
            }
        }

        // ---- epilogue: every wave has finished reading the activation buffer ----------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool is_stem = conv == 0;
        const bool is_conv2 = !is_stem && ((conv & 1) == 0);     // convs 1,3,5.. = conv1; 2,4,.. = conv2
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
                        lds + BIAS_OFF + (conv * CH + o0) * 4);
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
                    const int chunk = (o0 >> 3) ^ (p & 15);
                    *reinterpret_cast<__attribute__((address_space(3))) half4 *>(
                        lds + board * BOARD_BYTES + p * ROW_BYTES + (chunk << 4) + 8 * h) = o16;
                }
            }
        }
        // the barrier at the top of the next K-step orders these writes before the next reads
        if (VAR == 3) {                                 // ... unless activations are read before it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
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

    // ---- head 1x1 convolutions (policy: 2 channels, value: 1; BN folded) + ReLU, in fp32 -----------
    // model.py:40-42,51-55.  A position's 128 channels are spread over 2 waves x 2 lane halves:
    // each lane reduces its 32 channels, the 4 partial sums meet in LDS (the activation buffer is
    // dead now) and are added in a FIXED order (no float atomics: results are reproducible).
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // last epilogue's LDS traffic is over
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
