// tower_layer.hpp -- the split-precision trunk ("f16x3": operands carried as hi + lo fp16 pairs, products hi.Whi + lo.Whi
// + hi.Wlo; design of the arithmetic: tower_x16.hpp) at 256 filters as LAYER-WISE kernels (round 5).
//
// Replaces, at 256 filters, the trunk of ChessModel (/root/reference/src/chessrl/model.py:33-37,111-122: stem Conv3x3 + N x
// [Conv3x3-BN-ReLU-Conv3x3-BN-add-ReLU], BN folded on the host) where it has to be fp32-grade: the compliant precision modes
// of chessrl_amd/model.py (S2 of every simulation in "hybrid", everything in "f16x3").
//
// Why not the fused kernel.  k_trunk_x16 keeps the activations of its boards resident in LDS for the whole tower.  At 256
// filters x (hi, lo) a board is 66 KB: ONE board per workgroup, so every workgroup streams the whole weight set for one board --
// 171 B of weights through the LDS port per MFMA, 393 GB of L2 -> LDS traffic per launch at 20 x 256 and 4096 boards, 0.446 of
// the MFMA peak in issued FLOPs (profiles/r04).  Here the activations go back to HBM between layers (268 MB per layer and
// direction at 4096 boards: 0.1 ms at the chip's rate, spread under 0.55 ms of MFMA work) and a workgroup owns FOUR boards x all
// 256 output channels and streams BOTH operands:
//   * the input activations arrive per 32-channel K-chunk, [256 rows = 4 boards x 64 positions][hi 32 | lo 32] fp16 = 32 KiB,
//     by LDS-DMA from the global activation image into one of two LDS chunk buffers; all 9 taps of the chunk are served from it
//     (tap (dy,dx) = row p + 8dy + dx, off-board neighbours read a zero row with the same bank residue);
//   * the weights arrive as 16-KiB planes [256 rows][64 B] (Whi and Wlo of one (chunk, tap)) through a ring of four: 32 KiB of
//     weights per 768 MFMAs of the workgroup = 43 B per MFMA;
//   * a wave owns one board x 128 output channels: 4 x 8 accumulator blocks of 16 positions x 16 channels = 128 registers
//     (no skip stream in registers: it is the block's input, re-read from the activation image by the second convolution's
//     epilogue as hi + lo -- exact in fp32); per (chunk, tap) 4 + 4 activation and 8 + 8 weight fragment reads feed 96 MFMAs:
//     0.25 ds_read_b128 per MFMA (k_trunk_x16: 0.5);
//   * ONE barrier per (chunk, tap) = per 96 MFMAs of a wave: it sits where both weight planes of the tap are consumed into
//     registers, publishes the next tap's two planes (requested one tap earlier) and frees the two slots for the tap after;
//   * the epilogue writes hi / lo back to the global image in the layout the next layer's DMA wants:
//     [workgroup][chunk 8][row 256][hi 32 | lo 32]; the last layer also reduces the three 1x1 head convolutions.
// One launch per convolution (41 at 20 blocks) + one that expands the input planes: a kernel boundary costs ~2 us of a
// ~600-us layer, and it is the coherence point between a layer's stores and the next layer's loads; a persistent launch
// would have to hand-roll that and gains nothing (each workgroup's next layer depends on its own stores having landed).
// Measured (tools/ubench/conv_layer.hip, 4096 boards, one 256 -> 256 convolution incl. activation read + write): the loop
// alone 0.526 ms = 0.705 of the MFMA peak in issued FLOPs; first / second convolution of a block 0.563 / 0.606 ms = rocprofv3's
// 562.9 / 608.4 us inside a C5 bench (profiles/r05); k_trunk_x16<256, 1, SPLIT> took 0.82 ms per convolution.
// Accumulation order per output: chunk-major, tap, (hi.Whi, lo.Whi, hi.Wlo) -- NOT the order of k_trunk_x16 (tap-major): the
// library runs ONE arithmetic per (filters, mode), so at 256 filters every split-precision evaluation -- any batch size, the
// indexed fall-back launch of the hybrid mode too -- goes through these kernels (at four or at two boards per workgroup: the
// same accumulation order, the same bits).
#pragma once
#include "tower_x16.hpp"

// 1: inside the tap loop only waves 0 - 3 request the LDS-DMA transfers (twice the pieces each); their SIMD partners 4 - 7 go
// straight on with their MFMAs (see the tap body).  0: every wave requests its share behind the barrier.
#ifndef CRL_LAYER_DMA_HALF
#define CRL_LAYER_DMA_HALF 0
#endif
// 1 (the product since round 6): the MFMAs of the tap loop are inline asm with their accumulator tied in place.  Left to
// hipcc (0, rounds 5's product) some MFMAs get another destination than their C operand, the accumulators wander, and at four
// boards per workgroup 2 - 6 VGPRs end up in scratch (prologue values reloaded for the epilogue -- or, after an unrelated edit
// of the epilogue, whole accumulator blocks at the tail of the tap loop).  Tied in place the kernels take 232 - 236 VGPRs and
// no scratch at all, whatever the epilogue looks like (same MFMA sequence: the GPU suite's bit comparisons are unchanged).
#ifndef CRL_LAYER_ASM_MFMA
#define CRL_LAYER_ASM_MFMA 1
#endif

namespace crl_tower {

// NB boards per workgroup: 4 (full batches: a wave owns one board x 128 channels), 2 (batches of at most 512 boards: twice the
// workgroups, a wave owns one board x 64 channels -- 4 x 4 accumulator blocks) or 1 (eight waves x 32 channels of one board:
// the geometry for launches of a few hundred boards on 256 CUs, where the time of a launch is ONE workgroup's); every
// output is accumulated in the same order in all three: the same bits.
template <int NB_>
struct LayerGeoT {
    static_assert(NB_ == 4 || NB_ == 2 || NB_ == 1, "boards per workgroup");
    static constexpr int F = 256, NB = NB_, ROWS = NB * 64, TAPS = 9;
    static constexpr int CT = 2 * NB;                    // 16-channel blocks per wave: 8, 4, 2
    static constexpr int WPB = 8 / NB;                   // waves per board
    static constexpr int GROW = 128;                     // bytes of a row in the global image: hi 32 | lo 32 halves
    static constexpr int AROW = 160;                     // ... in LDS: + 32 B pad = 10 sixteen-byte units: the 16 lanes of a
                                                         // ds_read_b128 group (8 rows at quarter q, 8 at q + 1) cover all 64 banks
    static constexpr int ACHUNK = ROWS * AROW;           // 40 KiB (20 KiB)
    static constexpr int APIECES = NB == 4 ? 5 : (NB == 2 ? 3 : 2);   // 1-KiB DMA pieces per wave and chunk (NB 2: 24 pieces for
    static constexpr int ABUF = APIECES * 8 * 1024;      // 20 KiB, NB 1: 16 for 10 KiB; the surplus lands in a tail nobody reads)
    static constexpr int ZERO_OFF = 2 * ABUF;
    static constexpr int ZERO_BYTES = 16 * AROW;
    static constexpr int WRING_OFF = ((ZERO_OFF + ZERO_BYTES + 1023) / 1024) * 1024;
    static constexpr int TILE = F * 64;                  // one weight plane: [256 rows][64 B]
    // plane ring: pairs (Whi, Wlo) of a (chunk, tap).  Two pairs: the planes of tap u + 2 are requested at the barrier of tap u and
    // must have landed at the barrier of tap u + 1 -- one tap for an L2 / MALL round trip.  At four and two boards per workgroup a
    // tap is 1.3 / 0.65 us of MFMAs and that is enough; at ONE board it is 0.32 us against ~0.5 us of latency, and the launch
    // (the hybrid mode's indexed tower: its time is one workgroup's) ran at the latency, 0.53 - 0.64 us per tap.  There the ring
    // holds THREE pairs (LDS has the room): requested three taps ahead, two taps to land, the barrier's wait leaves the youngest
    // request in flight.  9 taps per chunk = 0 mod 3: the pair of a tap is t % 3, a compile-time number like the parity.
    static constexpr int WPAIRS = NB == 1 ? 3 : 2;
    static constexpr int LDS_BYTES = WRING_OFF + 2 * WPAIRS * TILE;
    static_assert(LDS_BYTES <= 160 * 1024 && ABUF >= ACHUNK, "LDS budget");
    static constexpr int CHUNK_BYTES = ROWS * GROW;      // one K-chunk of a workgroup's activations in the global image
    static constexpr int ACT_WG_BYTES = (F / 32) * CHUNK_BYTES;   // 64 KiB of (hi, lo) activations per board and layer
    static constexpr size_t conv_bytes(int chunks) { return (size_t)chunks * TAPS * 2 * TILE; }   // weight planes of one conv
};
typedef LayerGeoT<4> LayerGeo;

// IDX (k_layer_expand, k_layer_conv): 0 = the launch covers the batch.  Else it covers the boards of a device list (the hybrid
// mode's fall-back): 1 = whatever the list holds (harnesses); 2 = only a list of at most IDX_SMALL_MAX boards, 3 = only a longer
// one -- a launch whose list is outside its window leaves on its first instructions.  The list's length exists on the device
// only and the launch sequence of a captured hipGraph is fixed, so the product enqueues BOTH sequences behind one another -- one
// board per workgroup (IDX 2, a grid of IDX_SMALL_MAX workgroups: one round of the 256 CUs, where a launch takes ONE workgroup's
// time and that is 41 us per convolution instead of 65 at two boards) and two boards per workgroup (IDX 3: beyond one round
// the larger tile wins, 134 against 171 us at 800 boards) -- and the list picks the one that works; the idle sequence costs
// 41 x ~2 us (tools/ubench/conv_indexed.hip, profiles/r06/conv_indexed_harness.log).
constexpr int IDX_SMALL_MAX = 256;
template <int IDX>
__device__ __forceinline__ bool idx_outside(int listed)
{
    return (IDX == 2 && listed > IDX_SMALL_MAX) || (IDX == 3 && listed <= IDX_SMALL_MAX);
}

// uniform 64-bit base in SGPRs + unsigned 32-bit lane offset (through readfirstlane so that hipcc does not fold the lane
// offset into per-plane 64-bit lane addresses that it then hoists and spills)
__device__ __forceinline__ const unsigned char *uniform_ptr(const unsigned char *p)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const unsigned char *>(((unsigned long long)hi << 32) | lo);
}

// stage weight plane T of the stream (global image = LDS image, contiguous) into ring slot `slot`: 2 pieces of 1 KiB per
// wave; woff[j] = the lane's byte offset inside the plane for piece j
template <class G>
__device__ __forceinline__ void layer_stage_w(const unsigned char *wts, lds_byte *lds, int T, int slot, const unsigned (&woff)[2], int wave_u)
{
    const unsigned char *src = uniform_ptr(wts + (size_t)T * G::TILE);
    const int dst0 = G::WRING_OFF + slot * G::TILE + wave_u * 1024;
#pragma unroll
    for (int j = 0; j < 2; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + woff[j]),
                                         (__attribute__((address_space(3))) void *)(lds + dst0 + j * 8192), 16, 0, 0);
}

// piece j (0 .. APIECES - 1) of activation chunk `chunk` into LDS chunk buffer `buf`.  NB 4: wave w moves rows 32 w .. 32 w + 31
// (5 KiB of the padded image = 5 pieces); NB 2: the wave moves pieces w, w + 8, w + 16 of the 20-KiB image (pieces 20 .. 23
// re-read the last row into the buffer's tail).  Lanes that fall on a row's padding re-read its first 16 bytes.
template <class G>
__device__ __forceinline__ void layer_stage_act(const unsigned char *act, lds_byte *lds, int chunk, int buf, int j,
                                                unsigned voff, int wave_u)
{
    const unsigned char *src = uniform_ptr(act + (size_t)chunk * G::CHUNK_BYTES);
    const int dst = G::NB == 4 ? buf * G::ABUF + wave_u * 5120 + j * 1024 : buf * G::ABUF + (wave_u + 8 * j) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff),
                                     (__attribute__((address_space(3))) void *)(lds + dst), 16, 0, 0);
}

// In front of an indexed tower: every weight plane of the tower read once (one dword per 64-byte line; nothing is kept).  A list
// is a few dozen boards, so the 1 + 2 N convolutions behind it are latency-bound launches of a few dozen workgroups, each waiting
// one tap ahead for 32 KiB of planes that the S2 forward of the previous step (31 GB of activation traffic) has long flushed
// from the MALL: 2.5 ms per 41 convolutions at one board per workgroup with cold planes against 1.6 ms with hot ones.  Reading
// the 97 MB through all 256 CUs first takes ~30 us and leaves them in the memory-side cache: 1.86 ms in all
// (tools/ubench/conv_indexed.hip, profiles/r06/conv_indexed_harness_cold.log).  Leaves at once on an empty list.
__global__ __launch_bounds__(256) void k_layer_touch(const unsigned char *__restrict__ wts, size_t bytes, const int *__restrict__ list)
{
    if (list[0] <= 0) return;
    unsigned acc = 0;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 64; i < bytes; i += (size_t)gridDim.x * 256 * 64)
        acc += *reinterpret_cast<const unsigned *>(wts + i);
    asm volatile("" ::"v"(acc));
}

// The input planes of the listed / all boards as the first layer's activation image: [workgroup][chunk 4][row 256][hi 32 | lo
// 32], hi = the 0/1 plane values, lo = 0 (the stem then runs the same three products as every other layer; the one against
// lo adds exact zeros -- 1/123 of the tower's MFMAs for one kernel body less).
//   planes  BITS: 128 plane bitboards per board (u64 [n][128], bit sq of plane c = channel c on square sq; row 0 of the planes
//           is rank 8);  else fp16 [n][64][128]
//   list    IDX: int32 [LIST_HEADER + n], [0] = listed boards, [LIST_HEADER + k] = the board workgroup k / NB holds at slot k % NB
//           (a list that does not fill its last workgroup is padded with its last entry)
template <int BITS, int IDX, int NB = 4>
__global__ __launch_bounds__(512) void k_layer_expand(const unsigned char *__restrict__ planes, unsigned char *__restrict__ act_out,
                                                      const int *__restrict__ list)
{
    typedef LayerGeoT<NB> G;
    const int tid = threadIdx.x;
    int rows[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) rows[b] = blockIdx.x * NB + b;
    if constexpr (IDX) {
        const int listed = list[0];
        if ((int)blockIdx.x * NB >= listed || idx_outside<IDX>(listed)) return;
#pragma unroll
        for (int b = 0; b < NB; b++) rows[b] = list[LIST_HEADER + (rows[b] < listed ? rows[b] : listed - 1)];
    }
    unsigned char *out = act_out + (size_t)blockIdx.x * G::ACT_WG_BYTES;
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int item = k * 512 + tid;                 // (board, position, 16 channels)
        const int b = item >> 9, p = (item >> 3) & 63, c = item & 7;
        int row = rows[0];
#pragma unroll
        for (int bb = 1; bb < NB; bb++) row = b == bb ? rows[bb] : row;
        u32x4 v0, v1;
        if constexpr (BITS) {
            const int sq = p ^ 56;
            const unsigned long long *m = reinterpret_cast<const unsigned long long *>(planes) + (size_t)row * 128 + c * 16;
            unsigned int w[8];
#pragma unroll
            for (int e = 0; e < 8; e++)
                w[e] = (((m[2 * e] >> sq) & 1) ? 0x3C00u : 0u) | (((m[2 * e + 1] >> sq) & 1) ? 0x3C000000u : 0u);
            v0 = u32x4{w[0], w[1], w[2], w[3]};
            v1 = u32x4{w[4], w[5], w[6], w[7]};
        } else {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(planes + ((size_t)row * 64 + p) * 256 + c * 32);
            v0 = src[0];
            v1 = src[1];
        }
        unsigned char *dst = out + ((size_t)(c >> 1) * G::ROWS + b * 64 + p) * G::GROW + (c & 1) * 32;
        *reinterpret_cast<u32x4 *>(dst) = v0;
        *reinterpret_cast<u32x4 *>(dst + 16) = v1;
        *reinterpret_cast<u32x4 *>(dst + 64) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4 *>(dst + 80) = u32x4{0u, 0u, 0u, 0u};
    }
}

//   act_in   fp16 [n_wg][CHUNKS][256 rows][hi 32 | lo 32]      (row = 64 board + position, board 0..3 of the workgroup)
//   wts      fp16 planes of THIS convolution [chunk][tap 9][Whi, Wlo][256 rows][4 chunks][8]: rows / 16-byte chunks in the
//            LDS image's order (Geo16<256,...>::row_channel, wswz)
//   bias     f32 [256] of this convolution
//   act_out  [n_wg][8][256][hi 32 | lo 32]; KIND 2, 3: holds the block's input X on entry (the skip connection) and is
//            rewritten IN PLACE (a lane reads exactly the bytes it then writes)
// KIND 0: linear (the stem, model.py:33-34: no BN, no activation); 1: ReLU (first convolution of a block); 2: + X, ReLU
// (second); 3: as 2, and the tail of the tower: the three 1x1 head convolutions reduced to head_out f32 [n][192] (as
// k_trunk_x16 leaves them); 4: as 3, and the trunk's output to out f32 [n][64][256] (tests; its own instance so that the
// product's last layer carries neither the branch nor the registers).
// IDX != 0: the launch covers the list's boards (k_layer_expand; the windows of IDX 2 / 3: above); a workgroup beyond the list
// leaves at once; head_out rows by the list.
template <int CHUNKS, int KIND, int IDX, int NB = 4>
__global__ __launch_bounds__(512, 2) void k_layer_conv(const unsigned char *__restrict__ act_in,
                                                       const unsigned char *__restrict__ wts,
                                                       const float *__restrict__ bias,
                                                       unsigned char *__restrict__ act_out,
                                                       const int *__restrict__ list,
                                                       const float *__restrict__ head_w, const float *__restrict__ head_b,
                                                       float *__restrict__ head_out, float *__restrict__ out)
{
    typedef LayerGeoT<NB> G;
    typedef Geo16<256, 1, 1> WG;                        // weight plane order (row_channel, wswz, chan_of)
    static_assert(CHUNKS == 4 || CHUNKS == 8, "128 input planes or 256 channels");
    static_assert(NB == 4 || !(CRL_LAYER_DMA_HALF), "the loader-half experiment is written for four boards");
    constexpr int PT = 4, CT = G::CT, HC = CT / 2;      // position blocks, channel blocks, channel blocks per half
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    lds_byte *lds = (lds_byte *)lds_raw;
    const int lds_base = (int)(size_t)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int board = wave / G::WPB, obase = 16 * CT * (wave % G::WPB);
    const int r = lane & 15, q = lane >> 4;
    int listed = 0;
    if constexpr (IDX) {
        listed = __builtin_amdgcn_readfirstlane(list[0]);
        if ((int)blockIdx.x * NB >= listed || idx_outside<IDX>(listed)) return;    // before any DMA or barrier: the whole workgroup leaves
    }
    const unsigned char *act = act_in + (size_t)blockIdx.x * G::ACT_WG_BYTES;     // (the stem's image fills half a slot)

    // per-lane source offsets of the wave's activation pieces of a chunk (16-byte unit g of the padded image = row g / 10,
    // unit g % 10 of the row; units 8, 9 are padding)
    unsigned voff[G::APIECES];
#pragma unroll
    for (int j = 0; j < G::APIECES; j++) {
        const int g = NB == 4 ? wave * 320 + j * 64 + lane : (wave + 8 * j) * 64 + lane;
        const int row = g / 10 < G::ROWS ? g / 10 : G::ROWS - 1, col = g % 10;
        voff[j] = (unsigned)(row * G::GROW + (col < 8 ? col : 0) * 16);
    }

    // ---- prologue: chunk 0, the planes of taps 0 and 1, zero rows
#pragma unroll
    for (int j = 0; j < G::APIECES; j++) layer_stage_act<G>(act, lds, 0, 0, j, voff[j], wave_u);
    const unsigned woff[2] = {(unsigned)(tid * 16), (unsigned)(8192 + tid * 16)};
#pragma unroll
    for (int T = 0; T < 2 * G::WPAIRS; T++) layer_stage_w<G>(wts, lds, T, T, woff, wave_u);
    for (int i = tid; i < G::ZERO_BYTES / 16; i += 512)
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + G::ZERO_OFF + i * 16) = u32x4{0u, 0u, 0u, 0u};

    f32x4v acc[PT][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + obase + WG::chan_of(ct, 0) + 8 * q);
#pragma unroll
        for (int pt = 0; pt < PT; pt++) acc[pt][ct] = f32x4v{bv[0], bv[1], bv[2], bv[3]};
    }

    // the lane's activation row in block pt: 64 board + 16 pt + r; neighbours on the board as wave masks
    const int px = r & 7, py0 = r >> 3;
    const int base0 = lds_base + (board * 64 + r) * G::AROW + q * 16;
    const int zero_q = lds_base + G::ZERO_OFF + q * 16;
    const unsigned long long xm_left = __ballot(px >= 1), xm_right = __ballot(px <= 6);
    unsigned long long ym_up[PT], ym_down[PT];
#pragma unroll
    for (int pt = 0; pt < PT; pt++) {
        ym_up[pt] = __ballot(py0 + 2 * pt >= 1);
        ym_down[pt] = __ballot(py0 + 2 * pt <= 6);
    }
    auto tap_rows = [&](auto TAPC, int buf, int (&dst)[PT]) {
        constexpr int tap = decltype(TAPC)::value;
        constexpr int dy = tap / 3 - 1, dx = tap % 3 - 1;
        constexpr int shift = 8 * dy + dx;
        // (opaque per call: the addresses of all nine taps are loop invariants, and hipcc would otherwise compute the
        // 72 of them once in front of the chunk loop and spill them)
        int bb = base0 + buf * G::ABUF, rr = r;
        asm volatile("" : "+v"(bb), "+v"(rr));
        const int zrow = zero_q + ((rr + shift) & 15) * G::AROW;
        const int inb = bb + shift * G::AROW;
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            const int row = inb + pt * 16 * G::AROW;
            if constexpr (dx == 0 && dy == 0) {
                dst[pt] = row;
            } else {
                unsigned long long m;
                if constexpr (dy == 0) m = dx < 0 ? xm_left : xm_right;
                else if constexpr (dx == 0) m = dy < 0 ? ym_up[pt] : ym_down[pt];
                else m = (dx < 0 ? xm_left : xm_right) & (dy < 0 ? ym_up[pt] : ym_down[pt]);
                asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(dst[pt]) : "v"(zrow), "v"(row), "s"(m));
            }
        }
    };
    // weight fragment address: row obase + 16 ct + r of a plane, quarter q (the swizzle does not depend on ct)
    const int w0 = lds_base + G::WRING_OFF + (obase + r) * 64 + ((q ^ WG::wswz(obase + r)) << 4);
    const int w0hi = w0 + 4 * G::TILE;                  // (a third pair lies beyond the 16-bit offset of ds_read_b128)

    wait_vmcnt_n<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

#if defined(CRL_LAYER_STAMPS)
    // harness diagnostic (tools/ubench/conv_layer.hip -DCRL_LAYER_STAMPS): cycles a wave spends waiting for its DMA, at the
    // tap barrier, requesting DMA, in the whole loop and in the epilogue; written to `out` (never in the product)
    unsigned long long st_vm = 0, st_sb = 0, st_dma = 0, st_loop = 0, st_epi = 0, st_t0 = __builtin_amdgcn_s_memtime();
#endif
    half8 x[2][PT], w[2][HC];
    int ab[PT], abn[PT];

    // fragment reads (inline asm: counted by hand, see tower_common.hpp)
    auto rd_x = [&](half8 (&dst)[PT], const int (&rows)[PT], auto LO) {
        constexpr int lo = decltype(LO)::value;
#pragma unroll
        for (int pt = 0; pt < PT; pt++) dst[pt] = lds_read16_asm<lo * 64>(rows[pt]);
    };
    auto rd_w = [&](half8 (&dst)[HC], auto SLOT, auto HALF) {
        constexpr int slot = decltype(SLOT)::value, half = decltype(HALF)::value;
        static_for<0, HC>([&](auto CC) {
            constexpr int c = decltype(CC)::value;
            if constexpr (slot < 4) dst[c] = lds_read16_asm<slot * G::TILE + (half * HC + c) * 1024>(w0);
            else dst[c] = lds_read16_asm<(slot - 4) * G::TILE + (half * HC + c) * 1024>(w0hi);
        });
    };
    auto mfma16 = [&](const half8 (&ww)[HC], const half8 (&xx)[PT], auto HALF) {
        constexpr int half = decltype(HALF)::value;
#pragma unroll
        for (int pt = 0; pt < PT; pt++)
#pragma unroll
            for (int c = 0; c < HC; c++) {
#if CRL_LAYER_ASM_MFMA
                // accumulator tied in place ("+v"): hipcc otherwise gives some MFMAs another destination than their C
                // operand, moves accumulators around (100+ v_mov per chunk) and, as soon as the tap body holds a branch or a
                // second DMA site, spills dozens of them inside the loop.  Every hazard of these MFMAs is then ours: their
                // operands come from asm ds_reads behind counted waits; the first reader of the results is the epilogue,
                // behind the s_nops that follow the loop.
                f32x4v &a = acc[pt][half * HC + c];
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a) : "v"(ww[c]), "v"(xx[pt]));
#else
                acc[pt][half * HC + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ww[c], xx[pt], acc[pt][half * HC + c], 0, 0, 0);
#endif
            }
    };

    // cold start: hi rows of (chunk 0, tap 0) and Whi[0:4]
    tap_rows(I0{}, 0, ab);
    rd_x(x[0], ab, I0{});
    rd_w(w[0], I0{}, I0{});

    // One (chunk, tap): u = 9 c + t; parity P = u & 1 (compile time).  At its start the hi fragments sit in x[P] and Whi[0:4]
    // in w[P]; the Whi plane is in ring slot 2 P, Wlo in 2 P + 1, the next tap's planes in 2 (1 - P), 2 (1 - P) + 1.
    //        MFMAs (16 each)            fragment reads issued in front of them
    //   S1  hi x Whi[0:4]  w[P]        w[1-P] <- Whi[4:8]
    //   S2  hi x Whi[4:8]  w[1-P]      x[1-P] <- lo rows
    //   S3  lo x Whi[4:8]  w[1-P]
    //   S4  lo x Whi[0:4]  w[P]        w[1-P] <- Wlo[4:8]
    //   S5  hi x Wlo[4:8]  w[1-P]      w[P] <- Wlo[0:4];  x[1-P] <- the next tap's hi rows
    //   -- barrier: both planes of this tap are in registers; the next tap's planes (requested at the previous barrier)
    //      are published; the planes of the tap after next go into this tap's slots, + one piece of the next chunk
    //   S6  hi x Wlo[0:4]  w[P]        w[1-P] <- the next tap's Whi[0:4]
    // per output the order is hi.Whi, lo.Whi, hi.Wlo; the next tap starts with parity 1 - P.  Two fragment buffers of each
    // kind (64 registers) beside the 128 accumulators.
    auto tap_body = [&](auto PC, int c, auto TC) {
        constexpr int P = decltype(PC)::value, t = decltype(TC)::value;
        // ring pair of this tap: the parity with two pairs; t % 3 with three (9 taps per chunk: the same in every chunk)
        constexpr int R = G::WPAIRS == 2 ? P : t % 3;
        typedef std::integral_constant<int, 2 * R> SHI;
        typedef std::integral_constant<int, 2 * R + 1> SLO;
        typedef std::integral_constant<int, 2 * ((R + 1) % G::WPAIRS)> SNEXT;
        // LDS-DMA instructions this wave issued at the PREVIOUS tap's barrier (4 plane pieces + one activation piece in the
        // first APIECES taps of a chunk): with three pairs they may still be in flight at this tap's barrier
        constexpr int YOUNGEST = G::WPAIRS == 2 ? 0 : 4 + ((t >= 1 && t - 1 < G::APIECES) ? 1 : 0);
        const int u = c * G::TAPS + t;
        // (no branch in the body: behind the last tap the prefetches read valid LDS that nobody uses, and the DMA requests
        // re-load the last planes / the last chunk into slots and a buffer that are dead -- a branch around inline-asm
        // reads would invite phi copies of registers whose data has not landed, tower_x16.hpp)
        // S1
        rd_w(w[1 - P], SHI{}, I1{});
        wait_lgkm_n<HC>();                              // x[P] and w[P] have landed
        __builtin_amdgcn_sched_barrier(0);
        mfma16(w[P], x[P], I0{});
        __builtin_amdgcn_sched_barrier(0);
        // S2
        rd_x(x[1 - P], ab, I1{});
        wait_lgkm_n<PT>();                              // w[1-P] = Whi[4:8]
        __builtin_amdgcn_sched_barrier(0);
        mfma16(w[1 - P], x[P], I1{});
        __builtin_amdgcn_sched_barrier(0);
        // S3
        wait_lgkm_n<0>();                               // the lo rows
        __builtin_amdgcn_sched_barrier(0);
        mfma16(w[1 - P], x[1 - P], I1{});
        __builtin_amdgcn_sched_barrier(0);
        // S4 (+ the next tap's row addresses: VALU work beside the MFMAs)
        rd_w(w[1 - P], SLO{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mfma16(w[P], x[1 - P], I0{});
        {
            constexpr int tn = t + 1 == G::TAPS ? 0 : t + 1;
            const int cn = t + 1 == G::TAPS ? c + 1 : c;
            tap_rows(std::integral_constant<int, tn>{}, cn & 1, abn);
        }
        __builtin_amdgcn_sched_barrier(0);
        // S5
        rd_w(w[P], SLO{}, I0{});
        rd_x(x[1 - P], abn, I0{});
        wait_lgkm_n<HC + PT>();                         // w[1-P] = Wlo[4:8]
        __builtin_amdgcn_sched_barrier(0);
        mfma16(w[1 - P], x[P], I1{});
        __builtin_amdgcn_sched_barrier(0);
        // barrier: every fragment read of this tap's two planes has landed (only the next tap's x may be in flight)
        wait_lgkm_n<PT>();
#if defined(CRL_LAYER_STAMPS)
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
        wait_vmcnt_n<YOUNGEST>();
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long s2 = __builtin_amdgcn_s_memtime();
        st_vm += s1 - s0;
        st_sb += s2 - s1;
#else
        wait_vmcnt_n<YOUNGEST>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#endif
#if CRL_LAYER_DMA_HALF
        // The two waves of a SIMD (w and w + 4) leave the barrier together, and a request costs its wave ~115 cycles in which
        // it issues no MFMA (in-kernel stamps: 540 cycles per tap and wave, 11 % of the loop, both partners at the same
        // time).  Only waves 0 - 3 request -- all 32 plane pieces of the tap after next and, in the first five taps of a
        // chunk, two pieces each of the next chunk: their partners compute meanwhile and wait for them at the next barrier,
        // where the roles are reversed.
        if (wave_u < 4) {
            const int un = u + 2 < CHUNKS * G::TAPS ? u + 2 : CHUNKS * G::TAPS - 1;
            const unsigned char *src = uniform_ptr(wts + (size_t)(2 * un) * G::TILE);
#pragma unroll
            for (int j = 0; j < 8; j++)                 // two planes = 32 KiB contiguous in the stream: pieces 4 j + wave
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + (unsigned)(j * 4096) + woff[0]),
                    (__attribute__((address_space(3))) void *)(lds + G::WRING_OFF + (2 * P + (j >> 2)) * G::TILE + (j & 3) * 4096 + wave_u * 1024),
                    16, 0, 0);
            if constexpr (t < 5) {
                const int cn = c + 1 < CHUNKS ? c + 1 : CHUNKS - 1;
                const unsigned char *asrc = uniform_ptr(act + (size_t)cn * G::CHUNK_BYTES);
#pragma unroll
                for (int jj = 2 * t; jj < 2 * t + 2; jj++) {   // this wave: rows 64 w .. 64 w + 63 = 10 pieces of the padded image
                    int g = jj * 64 + lane;
                    asm volatile("" : "+v"(g));
                    const int rl = (g * 0xCCCD) >> 19, col = g - rl * 10;
                    const unsigned vo = (unsigned)((wave * 64 + rl) * G::GROW + (col < 8 ? col : 0) * 16);
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(asrc + vo),
                        (__attribute__((address_space(3))) void *)(lds + ((c + 1) & 1) * G::ACHUNK + wave_u * 10240 + jj * 1024), 16, 0, 0);
                }
            }
        }
#else
        {
            const int un = u + G::WPAIRS < CHUNKS * G::TAPS ? u + G::WPAIRS : CHUNKS * G::TAPS - 1;
            layer_stage_w<G>(wts, lds, 2 * un, 2 * R, woff, wave_u);
            layer_stage_w<G>(wts, lds, 2 * un + 1, 2 * R + 1, woff, wave_u);
        }
        if constexpr (t < G::APIECES) {
            // (the buffer of chunk c + 1 was last read in chunk c - 1, whose last barrier is behind us)
            const int cn = c + 1 < CHUNKS ? c + 1 : CHUNKS - 1;
            layer_stage_act<G>(act, lds, cn, (c + 1) & 1, t, voff[t < G::APIECES ? t : 0], wave_u);
        }
#endif
#if defined(CRL_LAYER_STAMPS)
        st_dma += __builtin_amdgcn_s_memtime() - s2;
#endif
        // S6
        rd_w(w[1 - P], SNEXT{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        mfma16(w[P], x[P], I0{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pt = 0; pt < PT; pt++) ab[pt] = abn[pt];
    };

    for (int c = 0; c < CHUNKS; c += 2) {
        static_for<0, G::TAPS>([&](auto TC) {
            constexpr int t = decltype(TC)::value;
            tap_body(std::integral_constant<int, t & 1>{}, c, TC);
        });
        static_for<0, G::TAPS>([&](auto TC) {
            constexpr int t = decltype(TC)::value;
            tap_body(std::integral_constant<int, (1 + t) & 1>{}, c + 1, TC);
        });
        // the reads in flight across the back-edge are drained: hipcc cannot see them and is free to copy their registers
        // there (phi moves) before the data has landed (tower_x16.hpp, tools/check_asm_hazards.py)
        wait_lgkm_n<0>();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if CRL_LAYER_ASM_MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last MFMAs' results are read below (16 passes of 4 cycles)
#endif
#if defined(CRL_LAYER_STAMPS)
    const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();
    st_loop = st_t1 - st_t0;
#endif

    // ---- epilogue: a lane holds, per (pt, g), the 8 consecutive channels obase + 32 g + 8 q .. + 7 of position 16 pt + r
    unsigned char *outp = act_out + (size_t)blockIdx.x * G::ACT_WG_BYTES;
    // KIND >= 3: the three 1x1 head convolutions.  A position's 256 channels live in 8 groups of 32 (one per (wave, g)) x 4 lane
    // quarters: 32 partial sums per output, parked in LDS and added in a fixed order that does not depend on the geometry
    // (no float atomics: reproducible, and the same bits at 4 and at 2 boards per workgroup).  The LDS is free once EVERY wave
    // is past its last fragment read: the barrier below.
    constexpr int NC = 32;
    __attribute__((address_space(3))) float *scratch = (__attribute__((address_space(3))) float *)lds;
    if constexpr (KIND >= 3) {
        // (opaque from here on: the head weights are loop invariants of the epilogue; hipcc would otherwise load all 96 of
        // a lane's values in front of the tap loop and spill accumulators to keep them)
        asm volatile("" : "+s"(head_w), "+s"(out));
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int g = 0; g < CT / 2; g++)
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
            // output chunk (obase + 32 g) / 32, row 64 board + 16 pt + r, quarter q
            unsigned char *dst = outp + ((size_t)((obase >> 5) + g) * G::ROWS + board * 64 + 16 * pt + r) * G::GROW + q * 16;
            f32x4v o[2] = {acc[pt][2 * g], acc[pt][2 * g + 1]};
            if constexpr (KIND >= 2) {
                // the skip connection: the block's input as it stands in the image, hi + lo (exact in fp32)
                const half8 xh = *reinterpret_cast<const half8 *>(dst), xl = *reinterpret_cast<const half8 *>(dst + 64);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    o[0][j] += (float)xh[j] + (float)xl[j];
                    o[1][j] += (float)xh[4 + j] + (float)xl[4 + j];
                }
            }
            if constexpr (KIND != 0) {
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int j = 0; j < 4; j++) o[h][j] = fmaxf(o[h][j], 0.f);
            }
            if constexpr (KIND >= 3) {
                const int o0 = obase + 32 * g + 8 * q;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const f32x4 wa = *reinterpret_cast<const f32x4 *>(head_w + k * G::F + o0);
                    const f32x4 wb = *reinterpret_cast<const f32x4 *>(head_w + k * G::F + o0 + 4);
                    float part = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) part += o[0][j] * wa[j];
#pragma unroll
                    for (int j = 0; j < 4; j++) part += o[1][j] * wb[j];
                    scratch[(((board * 64 + 16 * pt + r) * 3) + k) * NC + ((obase >> 5) + g) * 4 + q] = part;
                }
                if constexpr (KIND == 4) {
                    float *op = out + (((size_t)blockIdx.x * NB + board) * 64 + 16 * pt + r) * G::F + o0;
                    *reinterpret_cast<f32x4 *>(op) = f32x4{o[0][0], o[0][1], o[0][2], o[0][3]};
                    *reinterpret_cast<f32x4 *>(op + 4) = f32x4{o[1][0], o[1][1], o[1][2], o[1][3]};
                }
            } else {
                half8 hi8, lo8;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    hi8[j] = (_Float16)o[0][j];
                    hi8[4 + j] = (_Float16)o[1][j];
                    lo8[j] = (_Float16)(o[0][j] - (float)hi8[j]);
                    lo8[4 + j] = (_Float16)(o[1][j] - (float)hi8[4 + j]);
                }
                *reinterpret_cast<half8 *>(dst) = hi8;
                *reinterpret_cast<half8 *>(dst + 64) = lo8;
            }
        }
    if constexpr (KIND >= 3) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = tid; i < G::ROWS * 3; i += 512) {
            const int k = i % 3, bp = i / 3;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < NC; c++) v += scratch[i * NC + c];      // fixed order: channel group major, lane quarter minor
            v += head_b[k];
            size_t gb = (size_t)blockIdx.x * NB + (bp >> 6);
            if constexpr (IDX) {
                const int kk = (int)blockIdx.x * NB + (bp >> 6);
                gb = (size_t)list[LIST_HEADER + (kk < listed ? kk : listed - 1)];
            }
            const int pos = bp & 63;
            head_out[gb * 192 + (k < 2 ? pos * 2 + k : 128 + pos)] = fmaxf(v, 0.f);
        }
    }
#if defined(CRL_LAYER_STAMPS)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_epi = __builtin_amdgcn_s_memtime() - st_t1;
    if (lane == 0 && out) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(out) + ((size_t)blockIdx.x * 8 + wave) * 5;
        dbg[0] = st_loop; dbg[1] = st_epi; dbg[2] = st_vm; dbg[3] = st_sb; dbg[4] = st_dma;
    }
#endif
}

}  // namespace crl_tower
