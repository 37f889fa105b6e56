// heads.hpp -- the dense layers behind the tower's two head convolutions, on MFMA.
//
// Reference (paths relative to /root/reference/src/chessrl/): model.py:44-48 -- policy head
// Flatten -> Dense(1968, softmax); model.py:56-61 -- value head Flatten -> Dense(256, relu) ->
// Dense(1, tanh).  Inputs are the 192 head activations per board that the trunk kernel leaves
// (csrc/tower_x16.hpp: [0,128) policy, Keras Flatten order; [128,192) value).
//
// Round 1 ran these as three rocBLAS fp32 GEMMs + softmax + elementwise kernels (8 launches, 45-65 us
// per forward at 4096 boards, 20-35 us at 512: 4 % of a C3 step, 25 % of a C2 step).  Here: ONE
// launch per head.  fp32 accuracy comes from fp16 MFMAs on split operands: x = hi + lo with
// hi = fp16(x), lo = fp16(x - hi), and  W.h ~= Whi.hhi + Wlo.hhi + Whi.hlo  (the dropped lo.lo
// term is 2^-22 relative), accumulated in fp32 -- three v_mfma_f32_16x16x32_f16 per 16x16x32 block,
// 5x cheaper than v_mfma_f32_16x16x4_f32 (fp32 MFMA runs at the vector rate on gfx950).
//
// Orientation: D[output unit][board] = W^T . h^T, so a lane holds 4 consecutive output units of ONE
// board (lane = 16 q + r: board r of the 16-board block, units 4q..4q+3 of the 16-unit tile).  The
// weight operand is packed on the host in fragment order [tile][k-step][hi|lo][lane][8 halves] (one
// coalesced 1-KiB load per fragment, served by L2); the activation operand is split in-kernel.
#pragma once
#include "slices.hpp"
#include "tower_common.hpp"

namespace crl_heads {

using crl_tower::half8;
typedef float f32x4h __attribute__((ext_vector_type(4)));

constexpr int N_LABELS = 1968;
constexpr int N_LABELS_PAD = 2048;       // 8 waves x 16 tiles x 16 labels
constexpr int ACT = 192;                 // head activations per board

// activation fragments of k-step s for the 16-board block: lane (r, q) holds h[board0 + r][k0 + 32 s + 8 q ..+8]
__device__ __forceinline__ void split_act(const float *row, bool valid, int k, half8 &hi, half8 &lo)
{
    float x[8];
    if (valid) {
        const f32x4h a = *reinterpret_cast<const f32x4h *>(row + k);
        const f32x4h b = *reinterpret_cast<const f32x4h *>(row + k + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { x[e] = a[e]; x[4 + e] = b[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

// ---- value head: tanh(relu(h[64] . W1[64][256] + b1) . W2[256] + b2) -> value[n] -------------------
// One wavefront per 16 boards (block `vblock`).
__device__ __forceinline__ void value_head_block(const float *__restrict__ act, int n_boards, int vblock,
                                                 const unsigned char *__restrict__ w1p,    // packed fp16
                                                 const float *__restrict__ b1,
                                                 const float *__restrict__ w2,             // [257]: w2, then b2
                                                 float *__restrict__ value, int lane)
{
    const int r = lane & 15, q = lane >> 4;
    const int board = vblock * 16 + r;
    const bool valid = board < n_boards;
    const float *row = act + (size_t)board * ACT + 128;
    half8 hhi[2], hlo[2];
#pragma unroll
    for (int s = 0; s < 2; s++) split_act(row, valid, 32 * s + 8 * q, hhi[s], hlo[s]);
    const half8 *wf = reinterpret_cast<const half8 *>(w1p) + lane;
    float z = 0.f;
#pragma unroll
    for (int jt = 0; jt < 16; jt++) {
        f32x4h d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const half8 ahi = wf[((jt * 2 + s) * 2 + 0) * 64], alo = wf[((jt * 2 + s) * 2 + 1) * 64];
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, hhi[s], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, hhi[s], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, hlo[s], d, 0, 0, 0);
        }
        const f32x4h bv = *reinterpret_cast<const f32x4h *>(b1 + jt * 16 + 4 * q);
        const f32x4h wv = *reinterpret_cast<const f32x4h *>(w2 + jt * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; j++) z += fmaxf(d[j] + bv[j], 0.f) * wv[j];
    }
    z += __shfl_xor(z, 16);
    z += __shfl_xor(z, 32);
    if (valid && q == 0) value[board] = tanhf(z + w2[256]);    // b2 from memory: weights may change under a captured graph
}

__global__ __launch_bounds__(64) void k_value_head(const float *__restrict__ act, int n_boards,
                                                   const unsigned char *__restrict__ w1p,
                                                   const float *__restrict__ b1, const float *__restrict__ w2,
                                                   float *__restrict__ value)
{
    value_head_block(act, n_boards, blockIdx.x, w1p, b1, w2, value, threadIdx.x);
}

// ---- policy head: softmax(h[128] . W[128][1968] + b) -> policy[n][1968] ---------------------------
// One 512-thread workgroup per 16 NBLK boards; wave w owns labels [256 w, 256 w + 256) (16 tiles) of
// all its boards.  Every workgroup streams the whole packed kernel (1 MiB) from L2, which is what
// bounds the launch (256 workgroups x 1 MiB at ~10 TB/s): NBLK = 2 halves that traffic for large
// batches; NBLK = 1 keeps twice the workgroups for small ones.  Softmax statistics meet in LDS in a
// fixed order (no atomics: reproducible).
//
// LEGAL: the search kernels only ever read the policy at the labels of a position's legal moves
// (<= 218 of 1968).  Given that list (labels[board][256] in legal order, counts[board]) the kernel
// parks the exponentials of its 16 boards in LDS (128 KiB) and writes priors[board][j] =
// softmax[labels[board][j]] -- 4 x count bytes per board instead of 7 872, and the consumers read a
// contiguous row instead of gathering 4-byte words from a 7.9-KB one.  Same values, bit for bit
// (the same exp x (1 / sum) product).
constexpr int LEGAL_STRIDE = 256;                        // priors / labels per board (= CRL_MAX_MOVES)
constexpr int E_STRIDE = N_LABELS_PAD + 4;               // LDS row of a board: 16 rows start on 16 different bank quads
constexpr int LEGAL_LDS_BYTES = 16 * E_STRIDE * 4;

template <int NBLK, bool LEGAL = false>
__global__ __launch_bounds__(512, 2) void k_policy_head(const float *__restrict__ act, int n_boards,
                                                        const unsigned char *__restrict__ wp,   // packed fp16
                                                        const float *__restrict__ bias,          // [2048], pad = -1e30
                                                        float *__restrict__ policy,              // LEGAL: priors[n][256]
                                                        const unsigned short *__restrict__ labels = nullptr,
                                                        const int *__restrict__ counts = nullptr,
                                                        int pol_blocks = 0x7FFFFFFF,             // workgroups beyond: the value head
                                                        const unsigned char *__restrict__ w1p = nullptr,
                                                        const float *__restrict__ b1 = nullptr,
                                                        const float *__restrict__ w2 = nullptr,
                                                        float *__restrict__ value = nullptr)
{
    static_assert(!LEGAL || NBLK == 1, "the legal-move gather handles one 16-board block");
    __shared__ float s_max[NBLK][8][16], s_sum[NBLK][8][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= pol_blocks) {
        // the VALUE head rides in the same launch: workgroup pol_blocks + k serves board blocks 8 k .. 8 k + 7,
        // one wave each (k_value_head's arithmetic; no workgroup barrier on this path)
        const int vblock = ((int)blockIdx.x - pol_blocks) * 8 + wave;
        if (vblock * 16 < n_boards) value_head_block(act, n_boards, vblock, w1p, b1, w2, value, lane);
        return;
    }
    const int r = lane & 15, q = lane >> 4;
    int board[NBLK];
    bool valid[NBLK];
    half8 hhi[NBLK][4], hlo[NBLK][4];
#pragma unroll
    for (int nb = 0; nb < NBLK; nb++) {
        board[nb] = (blockIdx.x * NBLK + nb) * 16 + r;
        valid[nb] = board[nb] < n_boards;
        const float *row = act + (size_t)board[nb] * ACT;
#pragma unroll
        for (int s = 0; s < 4; s++) split_act(row, valid[nb], 32 * s + 8 * q, hhi[nb][s], hlo[nb][s]);
    }
    // weight fragments of this wave: tiles 16 wave .. 16 wave + 15, each 4 k-steps x (hi, lo) x 1 KiB
    const half8 *wf = reinterpret_cast<const half8 *>(wp) + (size_t)(wave * 16) * (4 * 2 * 64) + lane;
    f32x4h acc[NBLK][16];
    // the wave is bound by the latency of its weight loads (8 KiB per tile from L2), not by their
    // bandwidth: a ring of AHEAD + 1 tile buffers keeps AHEAD tiles in flight
    constexpr int AHEAD = NBLK == 1 ? 3 : 1;
    half8 a[AHEAD + 1][8];                               // [buffer][2 s + hl]
    f32x4h bvr[AHEAD + 1];                               // the tile's bias travels with its fragments:
    const float *bsrc = bias + wave * 256 + 4 * q;       // a load issued at its use would need vmcnt(0)
#pragma unroll
    for (int p = 0; p < AHEAD; p++) {
#pragma unroll
        for (int f = 0; f < 8; f++) a[p][f] = wf[(p * 8 + f) * 64];
        bvr[p] = *reinterpret_cast<const f32x4h *>(bsrc + p * 16);
    }
#pragma unroll
    for (int jt = 0; jt < 16; jt++) {
        const int cur = jt % (AHEAD + 1);
        if (jt + AHEAD < 16) {
#pragma unroll
            for (int f = 0; f < 8; f++) a[(jt + AHEAD) % (AHEAD + 1)][f] = wf[((jt + AHEAD) * 8 + f) * 64];
            bvr[(jt + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const f32x4h *>(bsrc + (jt + AHEAD) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the loads AHEAD tiles early (hipcc sinks them to their use)
        const f32x4h bv = bvr[cur];
        f32x4h d[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; nb++) d[nb] = f32x4h{bv[0], bv[1], bv[2], bv[3]};     // bias = C operand
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int nb = 0; nb < NBLK; nb++)
                d[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][2 * s], hhi[nb][s], d[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBLK; nb++)
                d[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][2 * s + 1], hhi[nb][s], d[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBLK; nb++)
                d[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][2 * s], hlo[nb][s], d[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NBLK; nb++) acc[nb][jt] = d[nb];
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- softmax over the 2048 (padded) labels of each board
#pragma unroll
    for (int nb = 0; nb < NBLK; nb++) {
        float m = acc[nb][0][0];
#pragma unroll
        for (int jt = 0; jt < 16; jt++)
#pragma unroll
            for (int j = 0; j < 4; j++) m = fmaxf(m, acc[nb][jt][j]);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (q == 0) s_max[nb][wave][r] = m;
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NBLK; nb++) {
        float mx = s_max[nb][0][r];
#pragma unroll
        for (int w = 1; w < 8; w++) mx = fmaxf(mx, s_max[nb][w][r]);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 16; jt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float e = __expf(acc[nb][jt][j] - mx);
                acc[nb][jt][j] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (q == 0) s_sum[nb][wave][r] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NBLK; nb++) {
        float tot = s_sum[nb][0][r];
#pragma unroll
        for (int w = 1; w < 8; w++) tot += s_sum[nb][w][r];
        const float inv = 1.0f / tot;
        if constexpr (LEGAL) {
            extern __shared__ __attribute__((aligned(16))) float s_e[];      // [16][E_STRIDE] + inv[16]
#pragma unroll
            for (int jt = 0; jt < 16; jt++)
                *reinterpret_cast<f32x4h *>(s_e + r * E_STRIDE + wave * 256 + jt * 16 + 4 * q) = acc[nb][jt];
            __syncthreads();
            const int b = tid >> 5, l = tid & 31;          // 32 threads per board
            const int gb = blockIdx.x * 16 + b;
            if (gb < n_boards) {
                // every lane of this half-wave needs board b's normaliser: lanes (r = b, q = 0) of
                // any wave hold it; recomputing it from s_sum is 8 LDS reads and the same bits
                float tb = s_sum[0][0][b];
#pragma unroll
                for (int w = 1; w < 8; w++) tb += s_sum[0][w][b];
                const float inv_b = 1.0f / tb;
                int cnt = counts[gb];
                cnt = cnt < 0 ? 0 : (cnt > LEGAL_STRIDE ? LEGAL_STRIDE : cnt);
                for (int j = l; j < cnt; j += 32) {
                    const int lab = labels[(size_t)gb * LEGAL_STRIDE + j] & (N_LABELS_PAD - 1);
                    policy[(size_t)gb * LEGAL_STRIDE + j] = s_e[b * E_STRIDE + lab] * inv_b;
                }
            }
        } else if (valid[nb]) {
            float *out = policy + (size_t)board[nb] * N_LABELS + wave * 256 + 4 * q;
#pragma unroll
            for (int jt = 0; jt < 16; jt++) {
                if (wave * 256 + jt * 16 < N_LABELS) {   // 1968 = 123 tiles: whole tiles only
                    f32x4h p;
#pragma unroll
                    for (int j = 0; j < 4; j++) p[j] = acc[nb][jt][j] * inv;
                    *reinterpret_cast<f32x4h *>(out + jt * 16) = p;
                }
            }
        }
    }
}

// ---- the same two heads for SMALL batches: label slices x board blocks ------------------------------
// k_policy_head gives every workgroup 16 boards and ALL 2048 labels: a batch of 512 boards is 32
// workgroups, each pulling the whole 1-MiB packed kernel through one CU (14 us: a fifth of a C2
// step, with 224 CUs idle).  Here the labels are cut into N_SLICES = 8 slices of 256: the grid is
// (9, board blocks) workgroups of 4 waves, slice s < 8 computing the logits of labels [256 s, +256)
// for its 16 boards (128 KiB of the packed kernel; a wave owns 4 tiles of 16 labels) and slice 8
// being the VALUE head of those boards (k_value_head's arithmetic, one wave), so the value no longer
// costs a launch of its own.  A softmax over slices needs two passes; pass 1 leaves
//     logits   (FULL: policy[board][label], LEGAL: priors[board][j] for the legal labels of the slice)
//     stats    [board][slice] = (m = max logit of the slice, s = sum exp(logit - m) over the slice)
// and pass 2 (k_policy_normalise) turns every stored logit l into  exp(l - M) / S  with
//     M = max_k m_k,   S = sum_k s_k exp(m_k - M)   (k = 0..7 in that order: reproducible).
// FULL and LEGAL run the same arithmetic on the same numbers: identical values, as in the one-pass
// kernel.  The values differ from the one-pass kernel's in the last bits (another summation order).
constexpr int N_SLICES = crl_slices::N_SLICES;
constexpr int SLICE_LABELS = N_LABELS_PAD / N_SLICES;     // 256
constexpr int SL_STRIDE = SLICE_LABELS + 4;               // LDS row of a board's slice logits

template <bool LEGAL>
__global__ __launch_bounds__(256) void k_heads_sliced(const float *__restrict__ act, int n_boards,
                                                      const unsigned char *__restrict__ wp,    // packed fp16 policy kernel
                                                      const float *__restrict__ bias,          // [2048], pad = -1e30
                                                      float *__restrict__ policy,              // logits out (FULL [n][1968] / LEGAL [n][256])
                                                      float2 *__restrict__ stats,              // [n][N_SLICES]
                                                      const unsigned short *__restrict__ labels,
                                                      const int *__restrict__ counts,
                                                      const unsigned char *__restrict__ w1p,   // value head (may be null: no value wanted)
                                                      const float *__restrict__ b1, const float *__restrict__ w2,
                                                      float *__restrict__ value)
{
    __shared__ float s_max[4][16], s_sum[4][16];
    __shared__ __attribute__((aligned(16))) float s_l[LEGAL ? 16 * SL_STRIDE : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int slice = blockIdx.x, board = blockIdx.y * 16 + r;
    const bool valid = board < n_boards;
    if (slice == N_SLICES) {
        // ---- value head of this board block: tanh(relu(h . W1 + b1) . W2 + b2), one wave
        if (wave != 0 || !value) return;
        value_head_block(act, n_boards, blockIdx.y, w1p, b1, w2, value, lane);
        return;
    }
    // ---- policy logits of labels [256 slice + 64 wave, +64) for the 16 boards
    half8 hhi[4], hlo[4];
    {
        const float *row = act + (size_t)board * ACT;
#pragma unroll
        for (int s = 0; s < 4; s++) split_act(row, valid, 32 * s + 8 * q, hhi[s], hlo[s]);
    }
    const int tile0 = slice * 16 + wave * 4;
    const half8 *wf = reinterpret_cast<const half8 *>(wp) + (size_t)tile0 * (4 * 2 * 64) + lane;
    half8 a[4][8];                                       // all four tiles in flight at once
    f32x4h acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int f = 0; f < 8; f++) a[t][f] = wf[(t * 8 + f) * 64];
        acc[t] = *reinterpret_cast<const f32x4h *>(bias + (tile0 + t) * 16 + 4 * q);      // bias = C operand
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][2 * s], hhi[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][2 * s + 1], hhi[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][2 * s], hlo[s], acc[t], 0, 0, 0);
        }
    // slice statistics of board r: max, then sum of exponentials, over 4 waves x 4 tiles x 16 labels
    float m = acc[0][0];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) m = fmaxf(m, acc[t][j]);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    if (q == 0) s_max[wave][r] = m;
    if constexpr (LEGAL) {
#pragma unroll
        for (int t = 0; t < 4; t++)
            *reinterpret_cast<f32x4h *>(s_l + r * SL_STRIDE + wave * 64 + t * 16 + 4 * q) = acc[t];
    }
    __syncthreads();
    const float mx = fmaxf(fmaxf(s_max[0][r], s_max[1][r]), fmaxf(s_max[2][r], s_max[3][r]));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) sum += __expf(acc[t][j] - mx);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (q == 0) s_sum[wave][r] = sum;
    __syncthreads();
    if (wave == 0 && q == 0 && valid)
        stats[(size_t)board * N_SLICES + slice] = make_float2(mx, ((s_sum[0][r] + s_sum[1][r]) + s_sum[2][r]) + s_sum[3][r]);
    if constexpr (LEGAL) {
        const int b = tid >> 4, l = tid & 15;            // 16 threads per board
        const int gb = blockIdx.y * 16 + b;
        if (gb < n_boards) {
            int cnt = counts[gb];
            cnt = cnt < 0 ? 0 : (cnt > LEGAL_STRIDE ? LEGAL_STRIDE : cnt);
            for (int j = l; j < cnt; j += 16) {
                const int lab = labels[(size_t)gb * LEGAL_STRIDE + j] & (N_LABELS_PAD - 1);
                if ((lab >> 8) == slice) policy[(size_t)gb * LEGAL_STRIDE + j] = s_l[b * SL_STRIDE + (lab & 255)];
            }
        }
    } else if (valid) {
        float *out = policy + (size_t)board * N_LABELS + tile0 * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < 4; t++)
            if ((tile0 + t) * 16 < N_LABELS)             // 1968 = 123 tiles: whole tiles only
                *reinterpret_cast<f32x4h *>(out + t * 16) = acc[t];
    }
}

// pass 2: every stored logit l of a board becomes exp(l - M) / S.  One wave per board.
template <bool LEGAL>
__global__ __launch_bounds__(256) void k_policy_normalise(float *__restrict__ policy, const float2 *__restrict__ stats,
                                                          int n_boards, const int *__restrict__ counts)
{
    const int lane = threadIdx.x & 63, board = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (board >= n_boards) return;
    const crl_slices::Norm nm = crl_slices::norm_of(stats + (size_t)board * N_SLICES);
    if constexpr (LEGAL) {
        int cnt = counts[board];
        cnt = cnt < 0 ? 0 : (cnt > LEGAL_STRIDE ? LEGAL_STRIDE : cnt);
        float *row = policy + (size_t)board * LEGAL_STRIDE;
        for (int j = lane; j < cnt; j += 64) row[j] = crl_slices::prob(row[j], nm);
    } else {
        f32x4h *row = reinterpret_cast<f32x4h *>(policy + (size_t)board * N_LABELS);     // 1968 floats = 492 quads, 16-byte aligned
        for (int j = lane; j < N_LABELS / 4; j += 64) {
            f32x4h v = row[j];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = crl_slices::prob(v[e], nm);
            row[j] = v;
        }
    }
}

// ---- hybrid precision: which S1 boards need the fp32-grade trunk? ------------------------------------
// An evaluation of S1 only chooses the opponent's reply: argmax of the policy over the legal labels
// (agentdistributed.py:57-58).  In the hybrid mode S1 runs the single-MFMA trunk first; this kernel lists the
// boards whose choice is not safe against that arithmetic's error: the log-margin of the two best legal
// moves, log p1 - log p2 (= the difference of their logits: a rounding error of the trunk moves a logit,
// i.e. a probability by a FACTOR), is below *log_margin -- read from DEVICE memory: the margin belongs to the
// weight set, weights are rewritten in place under captured hipGraphs, and a by-value kernel argument would
// stay what it was at capture.  One wave per board.  list: int32 [LIST_HEADER + n_boards]:
// [0] boards listed by this launch (zeroed by the caller), [1] unused, [2..3] one 64-bit running total of listed
// boards (statistics; an int32 wrapped after ~13 k C3 moves), [LIST_HEADER + k] the boards, in no particular
// order (the consumers do not depend on it).
// (the list's counter is zeroed by a kernel, not by hipMemsetAsync: a 4-byte memset node captured into a
// hipGraph did not run at replay on this stack -- the counter kept growing, the list overflowed)
__global__ __launch_bounds__(64) void k_zero_word(int *__restrict__ word)
{
    if (threadIdx.x == 0) *word = 0;
}

__global__ __launch_bounds__(256) void k_reply_margin(const float *__restrict__ priors, const int *__restrict__ counts,
                                                      int n_boards, const float *__restrict__ log_margin_p, int rows_are_logits,
                                                      int *__restrict__ list)
{
    const int lane = threadIdx.x & 63, board = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (board >= n_boards) return;
    int cnt = counts[board];
    cnt = cnt > LEGAL_STRIDE ? LEGAL_STRIDE : cnt;
    if (cnt < 2) return;                                  // no reply wanted, or a forced one
    const float *row = priors + (size_t)board * LEGAL_STRIDE;
    float m1 = -__builtin_inff(), m2 = -__builtin_inff();
    for (int j = lane; j < cnt; j += 64) {
        const float v = row[j];
        if (v > m1) { m2 = m1; m1 = v; } else if (v > m2) m2 = v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float o1 = __shfl_xor(m1, o), o2 = __shfl_xor(m2, o);
        const float hi = fmaxf(m1, o1), lo = fminf(m1, o1);
        m2 = fmaxf(lo, fmaxf(m2, o2));
        m1 = hi;
    }
    float margin;
    if (rows_are_logits) margin = m1 - m2;
    else margin = m2 > 0.f ? __logf(m1 / m2) : (m1 > 0.f ? __builtin_inff() : 0.f);
    if (lane == 0 && !(margin >= *log_margin_p)) {        // (NaN counts as unsafe)
        const int k = atomicAdd(list, 1);
        list[crl_tower::LIST_HEADER + k] = board;
        atomicAdd(reinterpret_cast<unsigned long long *>(list + 2), 1ull);
    }
}

}  // namespace crl_heads
