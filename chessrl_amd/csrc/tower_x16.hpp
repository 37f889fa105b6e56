// tower_x16.hpp -- the fused residual trunk (design: tower_common.hpp) on v_mfma_f32_16x16x32_f16
// (the round-1 kernels on v_mfma_f32_32x32x16_f16 left the tree in round 6; git history has them).
//
// Why: the trunk is matrix-pipe bound on a POWER-limited clock.  With the same 64x64 wave tile,
// the same LDS bytes per MAC and the same 64 accumulator registers, a 16x16x32 loop sustains
// 1.14x the FLOP/s of the 32x32x16 loop on MI355X (tools/ubench/mfma_shapes.hip: 1780 vs 1554
// TFLOP/s with every operand re-read from LDS; MI355X_MICROARCH.md, DVFS give-back item 7: equal
// cycles per FLOP, higher held clock).  Results are NOT bit-identical to the 32x32x16 kernels:
// the K = 32 contraction of one instruction sums in a different order (both accumulate fp32).
//
// What changes against the round-1 32x32x16 kernels:
//   * a wave tile of MT x NT 32x32 blocks becomes PT x CT = 2MT x 2NT blocks of 16 positions x 16
//     channels; a sub-step is 32 input channels: PT + CT fragment reads feed PT*CT MFMAs;
//   * fragment lanes: lane l = 16 q + r reads row r of its block, 16-byte quarter q of the 64-byte
//     K block.  Activation rows are padded to (2 mod 16) 16-byte units (288 / 544 bytes): the 16
//     lanes of a ds_read_b128 group -- 8 rows at quarter q, 8 at q+1 -- then cover all 64 banks;
//   * a weight tile is one 64-byte-row plane per 32 input channels, rows in the order the
//     accumulators want them (Geo16::row_channel), chunk swizzle (-(row >> 2)) & 3; the host packs
//     the global image in exactly that order, so the DMA source is contiguous;
//   * the accumulator of block (pt, ct) holds, per lane, 4 consecutive channels of position
//     16 pt + r; channel blocks 2g and 2g+1 are interleaved (Geo16::chan_of) so that a lane's 4 + 4
//     values are 8 consecutive channels and the epilogue writes 16-byte words in place;
//   * the head convolutions reduce 4 CG partial sums per output (4 lane quarters x CG waves).
#pragma once
#include "tower_common.hpp"

namespace crl_tower {

// SPLIT = 1 (precision mode "f16x3"): every activation is kept as TWO fp16 numbers, hi = fp16(x) and
// lo = fp16(x - hi), side by side in its LDS row ([hi: F channels | lo: F channels]), the folded
// weights likewise as two fp16 images, and a product is hi.Whi + hi.Wlo + lo.Whi (the dropped
// lo.Wlo term is 2^-22 relative): three MFMAs for fp32-grade results -- the same device as
// csrc/heads.hpp.  It is run as a convolution over an EXTENDED input: per spatial tap two weight "parts",
// the planes of Whi and the planes of Wlo, each a virtual tap like the channel halves of a 256-filter
// layer, so the tap loop, the weight ring and the pipeline are the ones below; Whi passes through the ring
// ONCE and every one of its weight sub-steps feeds two MFMA sub-steps, against hi and against lo (the
// stem's 0/1 planes have no lo: one); the weight image lists the planes in that order.
template <int F, int NB_, int SPLIT_ = 0>
struct Geo16 {
    static_assert(F == 64 || F == 128 || F == 256, "supported filter counts");
    static constexpr int SPLIT = SPLIT_;
    static constexpr int LO_OFF = F * 2;                // byte offset of the lo half inside an activation row
    static constexpr int NB = NB_;                      // boards per workgroup
    static constexpr int WPB = 8 / NB;                  // waves per board
    static constexpr int NT = (F == 64 && NB == 2) ? 1 : 2;
    static constexpr int CT = 2 * NT;                   // 16-channel blocks per wave
    static constexpr int CG = F / (16 * CT);            // channel groups (waves splitting the channels)
    static constexpr int PH = WPB / CG;                 // position halves (waves splitting a board)
    static constexpr int PT = 4 / PH;                   // 16-position blocks per wave
    static_assert(NB * CG * PH == 8 && (PH == 1 || PH == 2), "8 waves per workgroup");
    static constexpr int KT = F == 256 ? 32 : 64;       // input channels per weight tile
    static constexpr int SPT = KT / 32;                 // 32-channel sub-steps per tile
    static constexpr int WROW = KT * 2;
    static constexpr int WCH = WROW / 16;
    static constexpr int TILE_BYTES = F * WROW;
    static constexpr int GL = TILE_BYTES / 8192;
    static constexpr int AROW_CH = (SPLIT ? 2 * F : F) < 128 ? 128 : (SPLIT ? 2 * F : F);   // channels a row holds
    static constexpr int AROW = AROW_CH * 2 + 32;
    static constexpr int ABOARD = 64 * AROW;
    static constexpr int ZERO_OFF = NB * ABOARD;
    static constexpr int ZSTRIDE = AROW;                // 16 zero rows laid out like board rows (same bank residues)
    static constexpr int ZERO_BYTES = 16 * ZSTRIDE;
    static constexpr int BIAS_OFF = ZERO_OFF + ZERO_BYTES;          // float [2][F]: this layer's and the next one's
    static constexpr int WRING_OFF = ((BIAS_OFF + 2 * F * 4 + 1023) / 1024) * 1024;
    static constexpr int LDS_BYTES = WRING_OFF + PIPE_RING * TILE_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static constexpr int lds_bytes(int ring) { return WRING_OFF + ring * TILE_BYTES; }
    // LDS image of a weight tile: one plane per 32-channel sub-step, [SPT][F rows][64 bytes];
    // rows of 16 banks, chunk swizzle (-(row >> 2)) & 3: the 16 lanes of a ds_read_b128 group (8
    // rows at quarter q, 8 at q+1) cover all 64 banks, and the sub-step is an immediate offset.
    static constexpr int WPLANE = F * 64;
    __device__ static int wswz(int row) { return (0 - (row >> 2)) & 3; }
    // Which output channel accumulator row i (= 4 q + j) of channel block ct computes, relative to
    // the wave's first channel: blocks 2g and 2g+1 interleave in units of 4, so that a lane's 4 + 4
    // values of the two blocks are 8 CONSECUTIVE channels 32 g + 8 q .. + 8 and the epilogue writes
    // them as one 16-byte word (half the LDS write instructions, and conflict-free like the reads:
    // same (row, 16-byte quarter) lane map).  The permutation lives in the weight image (row_channel).
    __host__ __device__ static constexpr int chan_of(int ct, int i)
    {
        return 32 * (ct >> 1) + 8 * (i >> 2) + 4 * (ct & 1) + (i & 3);
    }
    // Output channel held by row `row` of a weight plane (rows 16 ct + i of a wave's 16 CT rows): the
    // same bit rotation inside every 32-row block whatever the geometry.  The GLOBAL weight image is
    // stored in exactly the LDS image's order -- planes of [F rows][64 bytes], rows in this order,
    // the four 16-byte chunks of a row at position chunk ^ wswz(row) (chessrl_amd/model.py:_pack_fused,
    // include/chessrl_hip.h) -- so a tile is one contiguous block that every wave copies in 1-KiB
    // pieces (sixteen 64-byte half lines per piece before: -0.8 % / -1.9 % at 128 / 256 filters).
    __host__ __device__ static constexpr int row_channel(int row)
    {
        return (row & ~31) + chan_of((row >> 4) & 1, row & 15);
    }
};

// stage weight tile t (global image = the plane image above, contiguous) into ring slot t & 3.
// (ALT 2: in-kernel cycle stamps; ALT 3, 4, 5, 6, 7: timing-only builds without the weight staging, without
// the per-tile barrier, without both, without the fragment reads from LDS, with NOTHING BUT the weight
// staging, barriers and epilogues (no fragment reads, no MFMAs) -- harness diagnostics, WRONG
// results, never dispatched.  ALT 8: correct results, round 2's schedule: the weight-fragment reads of the
// next sub-step issued in one clump before the MFMAs of half 1 instead of one by one between them.)
// ALT = 0: every wave moves GL pieces of 1 KiB.  ALT = 1: the tile is moved by ONE half of the
// workgroup -- waves 0-3 move even tiles, waves 4-7 odd tiles, 2 GL pieces each -- so that of the
// two waves sharing a SIMD only one sits in the LDS-DMA issue queue after a barrier while the other
// goes straight back to its MFMAs.
template <class G, int ALT>
__device__ inline void stage_wtile_x16(const unsigned char *wts, lds_byte *lds, int t, int tid, int wave_u,
                                       int slot = -1)
{
    const unsigned char *src = wts + (size_t)t * G::TILE_BYTES;
    const int slot0 = G::WRING_OFF + (slot < 0 ? (t & (PIPE_RING - 1)) : slot) * G::TILE_BYTES;
    if constexpr (ALT == 1) {
        if ((wave_u >> 2) != (t & 1)) return;
        const int dst0 = slot0 + (wave_u & 3) * 1024;   // uniform
#pragma unroll
        for (int j = 0; j < 2 * G::GL; j++) {
            const int idx = j * 256 + (tid & 255);      // 16-byte slot of the tile image (global = LDS order)
            const unsigned off = (unsigned)(idx * 16);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + off),
                (__attribute__((address_space(3))) void *)(lds + dst0 + j * 4096), 16, 0, 0);
        }
    } else {
        const int dst0 = slot0 + wave_u * 1024;         // uniform
#pragma unroll
        for (int j = 0; j < G::GL; j++) {
            const int idx = j * 512 + tid;              // 16-byte slot of the tile image (global = LDS order)
            const unsigned off = (unsigned)(idx * 16);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + off),
                (__attribute__((address_space(3))) void *)(lds + dst0 + j * 8192), 16, 0, 0);
        }
    }
}

// bias row of layer `conv` (F floats) -> LDS row conv & 1, by LDS-DMA: F/64 waves move 256 bytes each.
// Being a DMA like the weight tiles it is covered by the same counted vmcnt waits and barriers; no
// register, no ds_write and no compiler-inserted vmcnt(0) (which would drain the weight stream).
template <class G, int F>
__device__ inline void stage_bias_x16(const float *bias, lds_byte *lds, int conv, int lane, int wave_u)
{
    if (wave_u >= F / 64) return;
    // uniform base + unsigned 32-bit lane offset: the SGPR-base addressing form, no 64-bit lane address to keep
    // (the base goes through readfirstlane so that hipcc cannot fold the lane offset into a hoisted,
    // and then spilled, 64-bit per-lane address)
    const unsigned long long a = reinterpret_cast<unsigned long long>(bias + (size_t)conv * F + wave_u * 64);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned char *src = reinterpret_cast<const unsigned char *>(((unsigned long long)hi << 32) | lo);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(src + (unsigned)(lane * 4)),
        (__attribute__((address_space(3))) void *)(lds + G::BIAS_OFF + (conv & 1) * F * 4 + wave_u * 256), 4, 0, 0);
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt_n()
{
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkm_n()
{
    static_assert(N >= 0 && N < 16, "lgkmcnt immediate");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <int N> __device__ __forceinline__ void wait_lgkm()
{
    static_assert(N >= 0 && N <= 8 && N != 7, "lgkmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
}

//   planes  fp16 [n_boards][64][128], or 128 plane bitboards per board (BITS)
//   wts     fp16 weight planes in consumption order [conv][tap][in-ch/32][F rows][4 chunks][8 in],
//           rows and chunks in the LDS image's order (Geo16::row_channel, wswz)
//   bias    f32 [n_convs][F];  head_w f32 [3][F];  head_b f32 [3]
//   out     f32 [n_boards][64][F] or nullptr;  head_out f32 [n_boards][192] or nullptr
// PAIR = 1 (128 and 256 filters): ONE barrier per TWO weight tiles over a ring of five slots.  The
// sync sits at the start of the last sub-step of every odd tile t: tiles t+1, t+2 (moved at the
// previous sync) are published, tiles t+3, t+4 go into the slots of the dead tiles t-2, t-1.
// GROUP = 1 (64 filters).  There a wave tile is 2x2 / 2x4 blocks: a sub-step is 4 / 8 MFMAs per
// wave and a weight tile (one tap, 8 KiB) lasts 256 / 512 cycles.  The half-sub-step pipeline then
// leaves 2-4 MFMAs between a fragment read and its use (LDS latency is 10x that) and the ring of
// four looks one or two taps ahead of the L2 latency of the weight DMA: the loops ran at 30-45 % of
// their MFMA pace.  GROUP runs the taps in groups of three (one weight tile each, the stem's 128
// planes as two virtual taps of 64): ONE barrier per group, a weight ring of 3 (4 at two boards per
// workgroup) groups, so a tile is requested two (three) groups before it is read, and inside a
// group the six sub-steps run back to back with their fragments fetched TWO sub-steps ahead into
// three register buffers, across tap boundaries (not across layers: those activations do not exist yet).
// IDX = 1 (the fall-back launch of the hybrid precision mode, crl_trunk_forward_indexed): the launch covers a
// LIST of boards.  `out` is then not an output but the list, int32 [LIST_HEADER + n]: [0] = number of listed boards
// (a workgroup beyond it exits at once: the grid is sized for the worst case, the list is written on the
// device), [1 .. 3] unused here (statistics, csrc/heads.hpp), [LIST_HEADER + k] = the board (row of `planes` and of `head_out`) the k-th resident board
// is.  A list that does not fill the last workgroup is padded with its last entry (the same board computed
// twice writes the same values twice).  Nothing else changes: the production kernels (IDX = 0) compile as
// they did.
template <int F, int NB, int BITS = 0, int ALT = 0, int PAIR = 0, int GROUP = 0, int SPLIT = 0, int IDX = 0>
__global__ __launch_bounds__(512, 2) void k_trunk_x16(const unsigned char *__restrict__ planes,
                                                       const unsigned char *__restrict__ wts,
                                                       const float *__restrict__ bias,
                                                       float *__restrict__ out, int n_blocks,
                                                       const float *__restrict__ head_w,
                                                       const float *__restrict__ head_b,
                                                       float *__restrict__ head_out)
{
#if !defined(CRL_HARNESS)
    static_assert(ALT == 0, "diagnostic variants are for the harnesses under tools/ubench/ only");
#endif
    typedef Geo16<F, NB, SPLIT> G;
    static_assert(!SPLIT || (!GROUP && (ALT == 0 || ALT == 2)), "split precision runs the plain / pair pipelines");
    constexpr int PT = G::PT, CT = G::CT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    lds_byte *lds = (lds_byte *)lds_raw;
    const int lds_base = (int)(size_t)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int board = wave / G::WPB;
    const int obase = (16 * CT) * ((wave / G::PH) % G::CG);      // first output channel of this wave
    const int pbase = 32 * (wave % G::PH);              // first position of this wave
    const int r = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int n_convs = 1 + 2 * n_blocks;
    // SPLIT: two weight parts per tap (the planes of Whi, then of Wlo)
    const int tiles_stem = 9 * (SPLIT ? 2 : 1) * (128 / G::KT), tiles_conv = 9 * (SPLIT ? 2 : 1) * (F / G::KT);
    const int n_tiles = tiles_stem + 2 * n_blocks * tiles_conv;
    const size_t wg_board0 = (size_t)blockIdx.x * G::NB;
    int rows[G::NB];                                    // IDX: the listed boards this workgroup holds (wave-uniform)
    if constexpr (IDX) {
        static_assert(BITS == 1 && ALT == 0, "the indexed launch reads plane bitboards");
        const int *list = reinterpret_cast<const int *>(out);
        const int listed = __builtin_amdgcn_readfirstlane(list[0]);
        if ((int)wg_board0 >= listed) return;           // before any DMA or barrier: the whole workgroup leaves
#pragma unroll
        for (int b = 0; b < G::NB; b++) {
            const int k = (int)wg_board0 + b < listed ? (int)wg_board0 + b : listed - 1;
            rows[b] = __builtin_amdgcn_readfirstlane(list[LIST_HEADER + k]);
        }
    }

    static_assert(!PAIR || ((F == 128 || F == 256) && (ALT == 0 || ALT == 2 || ALT == 7 || ALT == 8)), "pair publishing: even tile counts per layer");
    static_assert(!SPLIT || ALT != 8, "the split-precision kernels run the production schedule");
    constexpr int GK = 3;                               // GROUP: taps (= tiles) per barrier
    constexpr int GRG = NB == 2 ? 4 : 3;                // GROUP: groups in the weight ring
    constexpr int GR = GK * GRG;                        // GROUP: ring slots
    static_assert(G::lds_bytes(GROUP ? GR : (PAIR ? 5 : PIPE_RING)) <= 160 * 1024, "LDS budget");
    static_assert(!GROUP || (F == 64 && !PAIR && (ALT == 0 || ALT == 2) && G::SPT == 2 && G::GL == 1 &&
                             2 * (PT + CT) < 16), "group pipeline: 64 filters");
    stage_bias_x16<G, F>(bias, lds, 0, lane, wave_u);   // oldest transfer: landed when tile 0 has
    if constexpr (GROUP) {
#pragma unroll
        for (int k = 0; k < GK * (GRG - 1); k++) stage_wtile_x16<G, 0>(wts, lds, k, tid, wave_u, k);   // groups 0 .. GRG-2
    } else {
        stage_wtile_x16<G, ALT>(wts, lds, 0, tid, wave_u, PAIR ? 0 : -1);
        stage_wtile_x16<G, ALT>(wts, lds, 1, tid, wave_u, PAIR ? 1 : -1);
        stage_wtile_x16<G, ALT>(wts, lds, 2, tid, wave_u, PAIR ? 2 : -1);
        if constexpr (PAIR) stage_wtile_x16<G, ALT>(wts, lds, 3, tid, wave_u, 3);
    }

    {   // planes (128 channels = 16 chunks per position) -> padded LDS rows; zero rows
        if constexpr (IDX) {
            expand_bitplanes_rows<G::NB, G::AROW, G::ABOARD>(planes, lds, rows, tid);
        } else if constexpr (BITS) {
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
    // tile 0 landed (tiles 1,2 may be in flight).  ALT: waves 0-3 moved tiles 0 and 2, waves 4-7 tile 1.
    // PAIR: tiles 0 AND 1 landed (1 is first read before the first sync), 2 and 3 in flight
    if constexpr (GROUP) wait_vmcnt_n<GK * (GRG - 2)>();              // group 0 landed, groups 1 .. GRG-2 in flight
    else if constexpr (ALT == 1) { if (wave_u < 4) wait_vmcnt<2 * G::GL>(); }
    else wait_vmcnt<2 * G::GL>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // position of the lane in block pt: p = pbase + 16 pt + r, i.e. file r & 7, rank (pbase >> 3) + 2 pt + (r >> 3)
    const int px = r & 7, py0 = (pbase >> 3) + (r >> 3);
    const int base0 = lds_base + board * G::ABOARD + (pbase + r) * G::AROW + q * 16;   // block 0's own row
    const int zero_q = lds_base + G::ZERO_OFF + q * 16;
    const int act_row0 = board * G::ABOARD + (pbase + r) * G::AROW;    // the lane's row in block 0
    // which lanes have an on-board neighbour to the left / right, and (per block) above / below
    const unsigned long long xm_left = __ballot(px >= 1), xm_right = __ballot(px <= 6);
    unsigned long long ym_up[PT], ym_down[PT];
#pragma unroll
    for (int pt = 0; pt < PT; pt++) {
        ym_up[pt] = __ballot(py0 + 2 * pt >= 1);
        ym_down[pt] = __ballot(py0 + 2 * pt <= 6);
    }
    // Weight fragment address: row o = obase + 16 ct + r of a tile plane, quarter q.  The swizzle
    // (-(o >> 2)) & 3 does not depend on ct (16 ct >> 2 is a multiple of 4), so ONE per-lane address
    // serves every channel block, sub-step and tile of a tap through the immediate offset
    // ct * 1024 + sub-step plane + tile slot; only the tap's first ring slot is added per tap.
    const int w0 = lds_base + G::WRING_OFF + (obase + r) * 64 + ((q ^ G::wswz(obase + r)) << 4);
    static_assert((16 * 64) == 1024 && G::TILE_BYTES * 2 + G::WPLANE + 3 * 1024 < 65536, "ds_read immediate");

    f32x4v res[PT][CT];                                 // fp32 residual stream
#pragma unroll
    for (int a = 0; a < PT; a++)
#pragma unroll
        for (int b = 0; b < CT; b++) res[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    constexpr int HP = PT / 2;                          // position blocks per half sub-step

    constexpr bool STAMP = ALT == 2;                    // harness diagnostic: in-kernel cycle stamps
    unsigned long long t_loop = 0, t_epi = 0, t_begin = 0, t_mark = 0, t_ba = 0, t_wr = 0, t_vm = 0, t_sb = 0;
    if constexpr (STAMP) { t_begin = __builtin_amdgcn_s_memtime(); t_mark = t_begin; }
    int t = 0;                                          // tile of the K-step being computed
    int slot_tap = 0;                                   // PAIR: ring slot (t mod 5) of the tap's first tile
    auto slot_add = [](int s, int k) { const int x = s + k; return x >= 5 ? x - 5 : x; };
    for (int conv = 0; conv < n_convs; conv++) {
        // the accumulators start from this layer's bias (LDS row conv & 1, staged one layer ahead):
        // the first MFMA of every chain takes it as its C operand, the epilogue adds nothing
        f32x4v acc[PT][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const f32x4 bv = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(
                lds + G::BIAS_OFF + (conv & 1) * F * 4 + (obase + G::chan_of(ct, 0) + 8 * q) * 4);
#pragma unroll
            for (int pt = 0; pt < PT; pt++) acc[pt][ct] = f32x4v{bv[0], bv[1], bv[2], bv[3]};
        }
        bool bias_staged = false;                       // next layer's bias goes out with the first tile sync

        // Fragment registers: a sub-step (32 channels) runs as two halves of PT/2 position blocks
        // against all CT channel blocks.  xa / xb: the activation fragments of the two halves;
        // w[2]: this sub-step's weight fragments and the next one's.  Half 0 prefetches xb (HP
        // reads); half 1 prefetches the next sub-step's xa and weights (HP + CT reads).
        half8 xa[HP], xb[HP], w[2][CT];
        int ab[2][PT];                                  // [current tap, next tap][pt]

        // one (virtual) tap = NS sub-steps of 32 input channels; sub-step i reads channel block i of
        // the tap's activation rows and sub-step i % SPT of weight tile t_tap0 + i / SPT
        // DUP = 2 (split precision, the Whi planes of a residual conv): every weight sub-step is used by
        // TWO MFMA sub-steps, first against the hi half of the activation rows, then against the lo
        // half LO_OFF bytes further -- the products hi.Whi and lo.Whi share one pass of Whi through the
        // ring and one set of weight fragment reads.  NS counts MFMA sub-steps, NS / DUP weight sub-steps.
        auto run_tap = [&](auto NSC, auto DUPC, bool first_tap, bool last_tap) {
            constexpr int NS = decltype(NSC)::value, DUP = decltype(DUPC)::value;
            static_assert(DUP == 1 || (SPLIT && DUP == 2 && ALT != 8), "shared weight sub-steps belong to the split kernels");
            const int t_tap0 = t;
            auto fetch_xa = [&](auto IC, bool next_tap) {
                constexpr int i = decltype(IC)::value;
                if constexpr (ALT == 7) return;
                if constexpr (ALT == 6) { if (t > 1) return; }       // timing only: no fragment reads
#pragma unroll
                for (int pt = 0; pt < HP; pt++)
                    xa[pt] = lds_read16_asm<(i / DUP) * 64 + (i % DUP) * G::LO_OFF>(ab[next_tap ? 1 : 0][pt]);
            };
            auto fetch_xb = [&](auto IC) {
                constexpr int i = decltype(IC)::value;
                if constexpr (ALT == 7) return;
                if constexpr (ALT == 6) { if (t > 1) return; }
#pragma unroll
                for (int pt = 0; pt < HP; pt++)
                    xb[pt] = lds_read16_asm<(i / DUP) * 64 + (i % DUP) * G::LO_OFF>(ab[0][HP + pt]);
            };
            // Ring of four: a tap's tiles sit in consecutive slots (a tap is 1, 2 or 4 tiles and
            // starts on a multiple of that), so slot and sub-step are immediates on top of the tap's
            // first slot.  Ring of five (PAIR): one base register per tile of the tap (slots wrap).
            constexpr int NSW = NS / DUP;                // weight sub-steps of the tap
            constexpr int TPT = NSW / G::SPT;
            static_assert((TPT == 1 || TPT == 2 || TPT == 4) && PIPE_RING == 4 && NSW % 2 == 0, "ring");
            int wv[PAIR ? TPT : 1], wv_nxt;
            if constexpr (PAIR) {
#pragma unroll
                for (int k = 0; k < TPT; k++) wv[k] = w0 + slot_add(slot_tap, k) * G::TILE_BYTES;
                wv_nxt = w0 + slot_add(slot_tap, TPT) * G::TILE_BYTES;
            } else {
                wv[0] = w0 + (t_tap0 & (PIPE_RING - 1)) * G::TILE_BYTES;
                wv_nxt = w0 + ((t_tap0 + TPT) & (PIPE_RING - 1)) * G::TILE_BYTES;
            }
            auto fetch_w1 = [&](auto IC, auto CC, bool next_tap, half8 (&dst)[CT]) {     // IC: weight sub-step
                constexpr int i = decltype(IC)::value, ct = decltype(CC)::value;
                if constexpr (ALT == 7) return;
                if constexpr (ALT == 6) { if (t > 1) return; }
                constexpr int off = ct * 1024 + (i % G::SPT) * G::WPLANE + (PAIR ? 0 : (i / G::SPT) * G::TILE_BYTES);
                dst[ct] = lds_read16_asm<off>(next_tap ? wv_nxt : wv[PAIR ? i / G::SPT : 0]);
            };
            auto fetch_w = [&](auto IC, bool next_tap, half8 (&dst)[CT]) {
                static_for<0, CT>([&](auto CC) { fetch_w1(IC, CC, next_tap, dst); });
            };
            if (first_tap) {
                fetch_xa(std::integral_constant<int, 0>{}, false);
                fetch_w(std::integral_constant<int, 0>{}, false, w[0]);
            }
            static_for<0, NS>([&](auto IC) {
                constexpr int i = decltype(IC)::value;
                constexpr int wi = i / DUP;              // weight sub-step of this MFMA sub-step
                constexpr bool w_first = i % DUP == 0, w_last = i % DUP == DUP - 1;
                constexpr int s = wi % G::SPT;
                constexpr int cur = wi % 2, nxt = 1 - cur;
                if constexpr (!w_first) {
                    // second use of this weight sub-step: no tile event
                } else if constexpr (PAIR) {
                    if constexpr (s == G::SPT - 1 && ((wi / G::SPT) & 1) == 1) {
                        // start of the last sub-step of an odd tile t: tiles t+1, t+2 (moved at the
                        // previous sync, the only transfers in flight) are published; tiles <= t-1
                        // are dead and their slots take tiles t+3, t+4
                        unsigned long long s0 = 0, s1 = 0;
                        if constexpr (STAMP) s0 = __builtin_amdgcn_s_memtime();
                        wait_vmcnt<0>();
                        if constexpr (STAMP) { s1 = __builtin_amdgcn_s_memtime(); t_vm += s1 - s0; }
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (STAMP) t_sb += __builtin_amdgcn_s_memtime() - s1;
                        if (!bias_staged) {
                            bias_staged = true;
                            if (conv + 1 < n_convs) stage_bias_x16<G, F>(bias, lds, conv + 1, lane, wave_u);
                        }
                        const int slot_cur = slot_add(slot_tap, wi / G::SPT);
                        if (t + 3 < n_tiles) stage_wtile_x16<G, 0>(wts, lds, t + 3, tid, wave_u, slot_add(slot_cur, 3));
                        if (t + 4 < n_tiles) stage_wtile_x16<G, 0>(wts, lds, t + 4, tid, wave_u, slot_add(slot_cur, 4));
                    }
                } else if constexpr (s == G::SPT - 1) {
                    // publish tile t+1 before the half that prefetches its first fragments;
                    // recycle tile t-1's slot
                    unsigned long long s0 = 0, s1 = 0;
                    if constexpr (STAMP) s0 = __builtin_amdgcn_s_memtime();
                    if constexpr (ALT == 1) {
                        // tile t+1 was moved by the half with (t+1) & 1, three syncs ago, and is
                        // the only transfer that half has in flight
                        if ((wave_u >> 2) == ((t + 1) & 1)) wait_vmcnt<0>();
                    } else if constexpr (ALT < 3) {
                        if (t + 2 < n_tiles) wait_vmcnt<G::GL>();
                        else wait_vmcnt<0>();
                    }
                    if constexpr (STAMP) { s1 = __builtin_amdgcn_s_memtime(); t_vm += s1 - s0; }
                    if constexpr (ALT != 4 && ALT != 5) __builtin_amdgcn_s_barrier();   // 4, 5: timing only
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (STAMP) t_sb += __builtin_amdgcn_s_memtime() - s1;
                    if (!bias_staged) {
                        bias_staged = true;
                        if (conv + 1 < n_convs) stage_bias_x16<G, F>(bias, lds, conv + 1, lane, wave_u);
                    }
                    if constexpr (ALT != 3 && ALT != 5)                                  // 3, 5: timing only
                        if (t + 3 < n_tiles) stage_wtile_x16<G, ALT>(wts, lds, t + 3, tid, wave_u);
                }
                // ---- half 0: position blocks [0, HP)
                fetch_xb(IC);
                wait_lgkm<HP>();                         // xa and w[cur] have landed
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ALT != 7) {
#pragma unroll
                for (int pt = 0; pt < HP; pt++)
#pragma unroll
                    for (int ct = 0; ct < CT; ct++)
                        acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[cur][ct], xa[pt], acc[pt][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- half 1: position blocks [HP, PT); prefetch the next sub-step
                constexpr bool wrap = i + 1 >= NS;
                bool issued = false;
                if constexpr (ALT != 8 && ALT != 7) {
                    // the next sub-step's activation fragments before the wait; its CT weight fragments
                    // one by one behind the first CT MFMAs of this half: a wave that issues all HP + CT
                    // reads in one clump keeps its MFMAs waiting behind 6 KiB of LDS transfers, -1.1 ..
                    // -1.6 % of kernel time at 128 and 256 filters (tools/ubench/trunk_r3.hip; ALT 8 = the
                    // clumped schedule of round 2; same MFMA order, bit-identical results)
                    if (!wrap || !last_tap) {
                        issued = true;
                        if constexpr (wrap) fetch_xa(std::integral_constant<int, 0>{}, true);
                        else fetch_xa(std::integral_constant<int, i + 1>{}, false);
                    }
                    if (!issued) wait_lgkm<0>();
                    else wait_lgkm<HP>();                // xb has landed
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, HP * CT>([&](auto KC) {
                        constexpr int k = decltype(KC)::value, pt = k / CT, ct = k % CT;
                        acc[HP + pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[cur][ct], xb[pt], acc[HP + pt][ct], 0, 0, 0);
                        if constexpr (k < CT && w_last) {                 // the next MFMA sub-step has new weights
                            __builtin_amdgcn_sched_barrier(0);
                            if (issued) {
                                if constexpr (wrap) fetch_w1(std::integral_constant<int, 0>{}, KC, true, w[nxt]);
                                else fetch_w1(std::integral_constant<int, wi + 1>{}, KC, false, w[nxt]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                if (!wrap || !last_tap) {
                    issued = true;
                    if constexpr (wrap) {
                        fetch_xa(std::integral_constant<int, 0>{}, true);
                        fetch_w(std::integral_constant<int, 0>{}, true, w[nxt]);
                    } else {
                        fetch_xa(std::integral_constant<int, i + 1>{}, false);
                        fetch_w(std::integral_constant<int, wi + 1>{}, false, w[nxt]);
                    }
                }
                if (!issued) wait_lgkm<0>();
                else wait_lgkm<HP + CT>();               // xb has landed
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ALT != 7) {
#pragma unroll
                for (int pt = 0; pt < HP; pt++)
#pragma unroll
                    for (int ct = 0; ct < CT; ct++)
                        acc[HP + pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[cur][ct], xb[pt], acc[HP + pt][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (s == G::SPT - 1 && w_last) t++;
            });
        };

        // A spatial tap with more than 128 input channels runs as Cin/128 "virtual taps" of 128
        // channels each (same neighbour rows, channel offset 256 bytes further): every virtual tap
        // is NS = 4 sub-steps, one unrolled body for the stem and the 128- and 256-filter layers.
        // (GROUP: virtual taps of 64 channels, so it is the stem's 128 planes that split in two)
        const int hshift = GROUP ? (conv == 0 ? 1 : 0) : ((conv != 0 && F == 256) ? 1 : 0);
        constexpr int chstep = GROUP ? 128 : 256;       // bytes between the channel halves of a spatial tap
        // SPLIT: a spatial tap is parts x pieces virtual taps (pieces = channel halves as above); part 0 =
        // the planes of Whi, used against hi and -- in the residual convs, whose input has a lo half -- in
        // the same pass against lo (run_tap DUP = 2); part 1 = the planes of Wlo against hi
        const int parts = SPLIT ? 2 : 1;
        const int per_tap = parts << hshift;
        const int nv = 9 * per_tap;
        // Activation row address of block pt for virtual tap v: the lane's own row shifted by
        // (8 dy + dx) rows when that neighbour is on the board, else the zero row whose index has
        // the same residue mod 16 (same banks).  The zero-row address and the file test do not
        // depend on pt: per block only the rank test and one select remain.
        auto vtap_rows = [&](int v, int (&dst)[PT]) {
            int tap, choff;
            if constexpr (SPLIT) {
                // order inside a spatial tap: part (Whi, Wlo) major, channel piece minor; the lo half of
                // the rows is an immediate of the reads (run_tap)
                tap = v / per_tap;
                const int rem = v - tap * per_tap;
                choff = (rem & ((1 << hshift) - 1)) * chstep;
            } else {
                tap = v >> hshift;
                choff = (v & ((1 << hshift) - 1)) * chstep;
            }
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int shift = 8 * dy + dx;
            const int zrow = zero_q + ((r + shift) & 15) * G::ZSTRIDE + choff;
            const int inb = base0 + shift * G::AROW + choff;
            // on-board tests as wave masks in SGPRs (computed once per kernel): per block only a
            // scalar AND, one add and one select remain
            const unsigned long long xm = dx < 0 ? xm_left : (dx > 0 ? xm_right : ~0ull);
#pragma unroll
            for (int pt = 0; pt < PT; pt++) {
                const unsigned long long m = xm & (dy < 0 ? ym_up[pt] : (dy > 0 ? ym_down[pt] : ~0ull));
                const int row = inb + pt * 16 * G::AROW;
                asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(dst[pt]) : "v"(zrow), "v"(row), "s"(m));
            }
        };
        if constexpr (GROUP) {
            constexpr int FR = PT + CT;                 // fragment reads of one sub-step
            half8 fx[3][PT], fw[3][CT];                 // sub-step i lives in buffer i % 3
            const int n_groups = n_tiles / GK;
            for (int g0 = 0; g0 < nv; g0 += GK) {       // virtual taps g0 .. g0+2 = tiles t .. t+2
                const bool more = g0 + GK < nv;         // the layer has another group
                int rows[GK + 1][PT], wb[GK + 1];
#pragma unroll
                for (int k = 0; k <= GK; k++) {
                    vtap_rows(g0 + k < nv ? g0 + k : 0, rows[k]);
                    int slot = slot_tap + k;
                    slot = slot >= GR ? slot - GR : slot;
                    wb[k] = w0 + slot * G::TILE_BYTES;
                }
                // sub-step j of the group (j >= 6: of the next group) -> buffer j % 3
                auto fetch = [&](auto JC) {
                    constexpr int j = decltype(JC)::value, b = j % 3, k = j / 2, pl = j % 2;
#pragma unroll
                    for (int pt = 0; pt < PT; pt++) fx[b][pt] = lds_read16_asm<pl * 64>(rows[k][pt]);
                    static_for<0, CT>([&](auto CC) {
                        constexpr int ct = decltype(CC)::value;
                        fw[b][ct] = lds_read16_asm<ct * 1024 + pl * G::WPLANE>(wb[k]);
                    });
                };
                if (g0 == 0) {                          // cold start of a layer
                    fetch(std::integral_constant<int, 0>{});
                    fetch(std::integral_constant<int, 1>{});
                }
                static_for<0, 2 * GK>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    if constexpr (i == 2 * GK - 2) {
                        // ---- the group's sync, before the first fetch from the next group's tiles:
                        // group tg+1 has landed everywhere, nobody reads group tg-1 any more -> its
                        // slots take group tg + GRG - 1
                        const int tg = t / GK;
                        if (tg + 1 < n_groups) {
                            unsigned long long s0 = 0, s1 = 0;
                            if constexpr (STAMP) s0 = __builtin_amdgcn_s_memtime();
                            if (tg + GRG - 2 < n_groups) wait_vmcnt_n<GK * (GRG - 3)>();   // younger groups stay in flight
                            else wait_vmcnt_n<0>();
                            if constexpr (STAMP) { s1 = __builtin_amdgcn_s_memtime(); t_vm += s1 - s0; }
                            __builtin_amdgcn_s_barrier();
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (STAMP) t_sb += __builtin_amdgcn_s_memtime() - s1;
                            if (!bias_staged) {
                                bias_staged = true;
                                if (conv + 1 < n_convs) stage_bias_x16<G, F>(bias, lds, conv + 1, lane, wave_u);
                            }
                            if (tg + GRG - 1 < n_groups) {
                                int slot = slot_tap - GK;                       // slots of group tg-1
                                slot = slot < 0 ? slot + GR : slot;
#pragma unroll
                                for (int k = 0; k < GK; k++)
                                    stage_wtile_x16<G, 0>(wts, lds, (tg + GRG - 1) * GK + k, tid, wave_u, slot + k);
                            }
                        }
                    }
                    // fetch sub-step i+2 (the last two reach into the next group of this layer)
                    if constexpr (i + 2 < 2 * GK) fetch(std::integral_constant<int, i + 2>{});
                    else if (more) fetch(std::integral_constant<int, i + 2>{});
                    // sub-step i's fragments have landed: at most the younger fetches stay outstanding
                    if constexpr (i + 2 < 2 * GK) wait_lgkm_n<2 * FR>();
                    else if constexpr (i == 2 * GK - 2) { if (more) wait_lgkm_n<2 * FR>(); else wait_lgkm_n<FR>(); }
                    else { if (more) wait_lgkm_n<2 * FR>(); else wait_lgkm_n<0>(); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pt = 0; pt < PT; pt++)
#pragma unroll
                        for (int ct = 0; ct < CT; ct++)
                            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i % 3][ct], fx[i % 3][pt], acc[pt][ct], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                t += GK;
                slot_tap += GK;
                slot_tap = slot_tap >= GR ? slot_tap - GR : slot_tap;
            }
        } else {
        vtap_rows(0, ab[1]);
        for (int v = 0; v < nv; v++) {
#pragma unroll
            for (int pt = 0; pt < PT; pt++) ab[0][pt] = ab[1][pt];
            vtap_rows(v + 1 < nv ? v + 1 : 0, ab[1]);
            constexpr std::integral_constant<int, 1> once{};
            if constexpr (SPLIT) {
                constexpr std::integral_constant<int, 2> twice{};
                const bool whi = ((v % per_tap) >> hshift) == 0;         // part 0
                if (conv != 0 && whi) {                                  // Whi against hi and lo
                    if (F == 64) run_tap(std::integral_constant<int, 4>{}, twice, v == 0, v == nv - 1);
                    else run_tap(std::integral_constant<int, 8>{}, twice, v == 0, v == nv - 1);
                } else if (F == 64 && conv != 0) run_tap(std::integral_constant<int, 2>{}, once, v == 0, v == nv - 1);
                else run_tap(std::integral_constant<int, 4>{}, once, v == 0, v == nv - 1);
            } else
            if (F == 64 && conv != 0) run_tap(std::integral_constant<int, 2>{}, once, v == 0, v == nv - 1);
            else run_tap(std::integral_constant<int, 4>{}, once, v == 0, v == nv - 1);
            if constexpr (PAIR) slot_tap = slot_add(slot_tap, 4 / G::SPT);
            // The fragments prefetched for the next tap are in flight across this loop's back-edge, where
            // hipcc -- which cannot see the inline-asm reads -- is free to COPY their registers (phi moves)
            // before the data has landed.  It does so in the split kernels, whose tap loop has several
            // bodies (found in the 64-filter ones as run-to-run differences once a second process delayed
            // the LDS returns); there the reads are drained here.  tools/check_asm_hazards.py walks the ISA of every trunk kernel for such reads and is
            // run by tests/test_host_and_cabi.py on every build.
            if constexpr (SPLIT) wait_lgkm<0>();
        }
        }

        // ---- epilogue ------------------------------------------------------------------------------
        if constexpr (STAMP) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_loop += now - t_mark; t_mark = now; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // all reads of the activation buffer done
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t_a = 0;
        if constexpr (STAMP) { t_a = __builtin_amdgcn_s_memtime(); t_ba += t_a - t_mark; }
        // three wave-uniform shapes (branches, not selects: every VALU instruction competes with the
        // partner wave's MFMAs for issue slots): stem = linear (no BN, no activation,
        // model.py:33-34); conv1 = ReLU, skip stream untouched; conv2 = + skip, ReLU, new skip
        const int kind = conv == 0 ? 0 : ((conv & 1) ? 1 : 2);
        // channel blocks 2g, 2g+1 of a position block: 8 consecutive channels per lane, one 16-byte write
        auto store_pair = [&](int pt, int g, const half4 &lo, const half4 &hi) {
            *reinterpret_cast<__attribute__((address_space(3))) half8 *>(
                lds + act_row0 + pt * 16 * G::AROW + (obase + 32 * g + 8 * q) * 2) =
                half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        const half4 zero4 = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
        // SPLIT: the value goes back as hi = fp16(v) and lo = fp16(v - hi), lo LO_OFF bytes further
        auto store_split = [&](int pt, int g, const f32x4v &a, const f32x4v &b) {
            half8 hi8, lo8;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                hi8[j] = (_Float16)a[j];
                hi8[4 + j] = (_Float16)b[j];
                lo8[j] = (_Float16)(a[j] - (float)hi8[j]);
                lo8[4 + j] = (_Float16)(b[j] - (float)hi8[4 + j]);
            }
            lds_byte *dst = lds + act_row0 + pt * 16 * G::AROW + (obase + 32 * g + 8 * q) * 2;
            *reinterpret_cast<__attribute__((address_space(3))) half8 *>(dst) = hi8;
            *reinterpret_cast<__attribute__((address_space(3))) half8 *>(dst + G::LO_OFF) = lo8;
        };
        if constexpr (SPLIT) {
#pragma unroll
            for (int g = 0; g < CT / 2; g++)
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
                    f32x4v o[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int ct = 2 * g + h;
                        if (kind == 0) {                // stem: linear, starts the skip stream
                            o[h] = acc[pt][ct];
                            res[pt][ct] = o[h];
                        } else if (kind == 1) {         // conv1: ReLU
#pragma unroll
                            for (int j = 0; j < 4; j++) o[h][j] = fmaxf(acc[pt][ct][j], 0.f);
                        } else {                        // conv2: + skip, ReLU, new skip
                            const f32x4v sum = acc[pt][ct] + res[pt][ct];
#pragma unroll
                            for (int j = 0; j < 4; j++) o[h][j] = fmaxf(sum[j], 0.f);
                            res[pt][ct] = o[h];
                        }
                    }
                    store_split(pt, g, o[0], o[1]);
                }
        } else
        if (kind == 1) {                                // conv1 of a block: ReLU, skip stream untouched
#pragma unroll
            for (int g = 0; g < CT / 2; g++)
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
                    half4 o16[2];                        // convert, then ReLU on packed halves
#pragma unroll
                    for (int h = 0; h < 2; h++) {
#pragma unroll
                        for (int j = 0; j < 4; j++) o16[h][j] = (_Float16)acc[pt][2 * g + h][j];
                        o16[h] = __builtin_elementwise_max(o16[h], zero4);
                    }
                    store_pair(pt, g, o16[0], o16[1]);
                }
        } else if (kind == 2) {                         // conv2: + skip, ReLU, new skip
#pragma unroll
            for (int g = 0; g < CT / 2; g++)
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
                    half4 o16[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int ct = 2 * g + h;
                        const f32x4v sum = acc[pt][ct] + res[pt][ct];      // packed fp32 adds
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float v = fmaxf(sum[j], 0.f);
                            res[pt][ct][j] = v;
                            o16[h][j] = (_Float16)v;
                        }
                    }
                    store_pair(pt, g, o16[0], o16[1]);
                }
        } else {                                        // stem: linear
#pragma unroll
            for (int g = 0; g < CT / 2; g++)
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
                    half4 o16[2];
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            res[pt][2 * g + h][j] = acc[pt][2 * g + h][j];
                            o16[h][j] = (_Float16)acc[pt][2 * g + h][j];
                        }
                    store_pair(pt, g, o16[0], o16[1]);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (STAMP) t_wr += __builtin_amdgcn_s_memtime() - t_a;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAMP) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_epi += now - t_mark; t_mark = now; }
    }
    if constexpr (STAMP) {
        // per wave: [loop cycles, epilogue cycles, epilogue phases, total, loop cycles waiting for weight DMA, at the tile barrier]
        if (lane == 0 && out) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(out) + ((size_t)blockIdx.x * 8 + wave) * 6;
            dbg[0] = t_loop; dbg[1] = t_epi; dbg[2] = (t_ba << 32) | (t_wr & 0xffffffffull); dbg[3] = __builtin_amdgcn_s_memtime() - t_begin;
            dbg[4] = t_vm; dbg[5] = t_sb;
        }
        return;
    }

    if (!IDX && out) {
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            const int p = pbase + 16 * pt + r;
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const int o0 = obase + G::chan_of(ct, 0) + 8 * q;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = res[pt][ct][j];
                *reinterpret_cast<f32x4 *>(out + ((wg_board0 + board) * 64 + p) * F + o0) = v;
            }
        }
    }

    if (head_out) {
        // a position's F channels live in CG waves x 4 lane quarters: 4*CG partial sums per output,
        // added in a fixed order (no float atomics: results are reproducible)
        constexpr int NC = 4 * G::CG;
        float part[PT][3];
#pragma unroll
        for (int pt = 0; pt < PT; pt++)
#pragma unroll
            for (int k = 0; k < 3; k++) part[pt][k] = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const int o0 = obase + G::chan_of(ct, 0) + 8 * q;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(head_w + k * F + o0);
#pragma unroll
                for (int pt = 0; pt < PT; pt++)
#pragma unroll
                    for (int j = 0; j < 4; j++) part[pt][k] += res[pt][ct][j] * wv[j];
            }
        }
        __attribute__((address_space(3))) float *scratch = (__attribute__((address_space(3))) float *)lds;
#pragma unroll
        for (int pt = 0; pt < PT; pt++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                scratch[(((board * 64 + pbase + 16 * pt + r) * 3) + k) * NC + (obase / (16 * CT)) * 4 + q] = part[pt][k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = tid; i < G::NB * 64 * 3; i += 512) {
            const int k = i % 3, bp = i / 3;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < NC; c++) v += scratch[i * NC + c];      // fixed order
            v += head_b[k];
            size_t gb = wg_board0 + (bp >> 6);
            if constexpr (IDX) {
                gb = (size_t)rows[0];
#pragma unroll
                for (int b = 1; b < G::NB; b++) gb = (bp >> 6) == b ? (size_t)rows[b] : gb;
            }
            const int pos = bp & 63;
            head_out[gb * 192 + (k < 2 ? pos * 2 + k : 128 + pos)] = fmaxf(v, 0.f);
        }
    }
}

}  // namespace crl_tower
