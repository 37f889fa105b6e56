// Scratch (GPU), round 5: would ONE launch per layer-wise tower pay?  k_layer_tower below is k_layer_conv's convolution
// (chessrl_amd/csrc/tower_layer.hpp: same tap body, same accumulation order, same epilogues) inside a loop over the
// layers of the tower: a workgroup runs all 1 + 2 x blocks convolutions of its four boards back to back -- each layer reads
// only what the SAME workgroup wrote (vmcnt(0) + s_barrier between its stores and its next LDS-DMA) -- so the 41 grid-wide
// hand-overs of the launch sequence disappear, the next layer's first weight planes are requested in front of the epilogue,
// and the workgroups of a round drift apart (their epilogues, which run at the chip's memory rate when all 256 of a round
// write at once, no longer coincide).  Compared here with the product's launch sequence on the same buffers: head outputs
// must be bit-identical.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I chessrl_amd/csrc tools/ubench/tower_persist.hip -o tools/ubench/tower_persist
//   ./tower_persist [boards=4096] [blocks=20] [reps=5]
#define CRL_HARNESS 1
#include "tower_layer.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace crl_tower {

//   img_a, img_b  the two activation images [n_wg][8][ROWS][hi 32 | lo 32]; img_b holds the expanded input planes (4 chunks)
//                 on entry (k_layer_expand); stem: b -> a, first convolution of a block: a -> b, second: b -> a in place
//   wts           the planes of all convolutions (stem: 4 chunks, then 8 chunks each), bias f32 [1 + 2 n_blocks][256]
template <int IDX, int NB = 4>
__global__ __launch_bounds__(512, 2) void k_layer_tower(unsigned char *__restrict__ img_a, unsigned char *__restrict__ img_b,
                                                        const unsigned char *__restrict__ wts, const float *__restrict__ bias,
                                                        int n_blocks, const int *__restrict__ list,
                                                        const float *__restrict__ head_w, const float *__restrict__ head_b,
                                                        float *__restrict__ head_out, int stagger_ticks)
{
    typedef LayerGeoT<NB> G;
    // round 6: the workgroups of a launch all do the same work, so without help they stay in lockstep and all 256 of a round
    // reach their epilogue -- 256 KiB of stores (+ 256 KiB of skip rows) each, at the chip's memory rate -- at the same time.
    // The first resident round is started in four phases, one per XCD pair (blockIdx & 3 = XCD & 3: the workgroups of an XCD
    // stay together and keep sharing their weight planes in its L2); every later workgroup inherits the phase of the one it
    // follows on its CU.  s_memrealtime: 100 MHz.
    if (stagger_ticks > 0 && blockIdx.x < 256) {
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(blockIdx.x & 3) * stagger_ticks;
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(32);
    }
    typedef Geo16<256, 1, 1> WG;
    constexpr int PT = 4, CT = G::CT, HC = CT / 2;
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
        if ((int)blockIdx.x * NB >= listed) return;
    }
    unsigned char *wg_a = img_a + (size_t)blockIdx.x * G::ACT_WG_BYTES, *wg_b = img_b + (size_t)blockIdx.x * G::ACT_WG_BYTES;

    unsigned voff[G::APIECES];
#pragma unroll
    for (int j = 0; j < G::APIECES; j++) {
        const int g = NB == 4 ? wave * 320 + j * 64 + lane : (wave + 8 * j) * 64 + lane;
        const int row = g / 10 < G::ROWS ? g / 10 : G::ROWS - 1, col = g % 10;
        voff[j] = (unsigned)(row * G::GROW + (col < 8 ? col : 0) * 16);
    }
    const unsigned woff[2] = {(unsigned)(tid * 16), (unsigned)(8192 + tid * 16)};
    for (int i = tid; i < G::ZERO_BYTES / 16; i += 512)
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + G::ZERO_OFF + i * 16) = u32x4{0u, 0u, 0u, 0u};

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
    const int w0 = lds_base + G::WRING_OFF + (obase + r) * 64 + ((q ^ WG::wswz(obase + r)) << 4);

    half8 x[2][PT], w[2][HC];
    int ab[PT], abn[PT];
    f32x4v acc[PT][CT];
    auto rd_x = [&](half8 (&dst)[PT], const int (&rows)[PT], auto LO) {
        constexpr int lo = decltype(LO)::value;
#pragma unroll
        for (int pt = 0; pt < PT; pt++) dst[pt] = lds_read16_asm<lo * 64>(rows[pt]);
    };
    auto rd_w = [&](half8 (&dst)[HC], auto SLOT, auto HALF) {
        constexpr int slot = decltype(SLOT)::value, half = decltype(HALF)::value;
        static_for<0, HC>([&](auto CC) {
            constexpr int c = decltype(CC)::value;
            dst[c] = lds_read16_asm<slot * G::TILE + (half * HC + c) * 1024>(w0);
        });
    };
    auto mfma16 = [&](const half8 (&ww)[HC], const half8 (&xx)[PT], auto HALF) {
        constexpr int half = decltype(HALF)::value;
#pragma unroll
        for (int pt = 0; pt < PT; pt++)
#pragma unroll
            for (int c = 0; c < HC; c++)
                acc[pt][half * HC + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ww[c], xx[pt], acc[pt][half * HC + c], 0, 0, 0);
    };

    const int n_layers = 1 + 2 * n_blocks;
    const unsigned char *lw = wts;                     // this layer's planes
    // the first layer's first four planes (every later layer's are requested in front of the epilogue before it)
#pragma unroll
    for (int T = 0; T < 4; T++) layer_stage_w<G>(lw, lds, T, T, woff, wave_u);

    for (int L = 0; L < n_layers; L++) {
        const int chunks = L == 0 ? 4 : 8;
        const int kind = L == 0 ? 0 : ((L & 1) ? 1 : (L + 1 == n_layers ? 3 : 2));
        const unsigned char *act = (L == 0 || !(L & 1)) ? wg_b : wg_a;          // stem and second convolutions read b
        unsigned char *outp = (L == 0 || !(L & 1)) ? wg_a : wg_b;
        const float *lb = bias + (size_t)L * G::F;
        asm volatile("" : "+s"(act), "+s"(outp), "+s"(lb), "+s"(lw));

        // ---- prologue: chunk 0 (what this workgroup's previous layer stored: behind its vmcnt(0) + barrier)
#pragma unroll
        for (int j = 0; j < G::APIECES; j++) layer_stage_act<G>(act, lds, 0, 0, j, voff[j], wave_u);
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lb + obase + WG::chan_of(ct, 0) + 8 * q);
#pragma unroll
            for (int pt = 0; pt < PT; pt++) acc[pt][ct] = f32x4v{bv[0], bv[1], bv[2], bv[3]};
        }
        wait_vmcnt_n<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        tap_rows(I0{}, 0, ab);
        rd_x(x[0], ab, I0{});
        rd_w(w[0], I0{}, I0{});

        auto tap_body = [&](auto PC, int c, auto TC) {
            constexpr int P = decltype(PC)::value, t = decltype(TC)::value;
            typedef std::integral_constant<int, 2 * P> SHI;
            typedef std::integral_constant<int, 2 * P + 1> SLO;
            typedef std::integral_constant<int, 2 * (1 - P)> SNEXT;
            const int u = c * G::TAPS + t;
            rd_w(w[1 - P], SHI{}, I1{});
            wait_lgkm_n<HC>();
            __builtin_amdgcn_sched_barrier(0);
            mfma16(w[P], x[P], I0{});
            __builtin_amdgcn_sched_barrier(0);
            rd_x(x[1 - P], ab, I1{});
            wait_lgkm_n<PT>();
            __builtin_amdgcn_sched_barrier(0);
            mfma16(w[1 - P], x[P], I1{});
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkm_n<0>();
            __builtin_amdgcn_sched_barrier(0);
            mfma16(w[1 - P], x[1 - P], I1{});
            __builtin_amdgcn_sched_barrier(0);
            rd_w(w[1 - P], SLO{}, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mfma16(w[P], x[1 - P], I0{});
            {
                constexpr int tn = t + 1 == G::TAPS ? 0 : t + 1;
                const int cn = t + 1 == G::TAPS ? c + 1 : c;
                tap_rows(std::integral_constant<int, tn>{}, cn & 1, abn);
            }
            __builtin_amdgcn_sched_barrier(0);
            rd_w(w[P], SLO{}, I0{});
            rd_x(x[1 - P], abn, I0{});
            wait_lgkm_n<HC + PT>();
            __builtin_amdgcn_sched_barrier(0);
            mfma16(w[1 - P], x[P], I1{});
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkm_n<PT>();
            wait_vmcnt_n<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            {
                const int un = u + 2 < chunks * G::TAPS ? u + 2 : chunks * G::TAPS - 1;
                layer_stage_w<G>(lw, lds, 2 * un, 2 * P, woff, wave_u);
                layer_stage_w<G>(lw, lds, 2 * un + 1, 2 * P + 1, woff, wave_u);
            }
            if constexpr (t < G::APIECES) {
                const int cn = c + 1 < chunks ? c + 1 : chunks - 1;
                layer_stage_act<G>(act, lds, cn, (c + 1) & 1, t, voff[t < G::APIECES ? t : 0], wave_u);
            }
            rd_w(w[1 - P], SNEXT{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            mfma16(w[P], x[P], I0{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pt = 0; pt < PT; pt++) ab[pt] = abn[pt];
        };

        for (int c = 0; c < chunks; c += 2) {
            static_for<0, G::TAPS>([&](auto TC) {
                constexpr int t = decltype(TC)::value;
                tap_body(std::integral_constant<int, t & 1>{}, c, TC);
            });
            static_for<0, G::TAPS>([&](auto TC) {
                constexpr int t = decltype(TC)::value;
                tap_body(std::integral_constant<int, (1 + t) & 1>{}, c + 1, TC);
            });
            wait_lgkm_n<0>();
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // every wave is past its last fragment read: the ring and the chunk buffers are free
        __builtin_amdgcn_s_barrier();
        lw += G::conv_bytes(chunks);
        if (L + 1 < n_layers) {
#pragma unroll
            for (int T = 0; T < 4; T++) layer_stage_w<G>(lw, lds, T, T, woff, wave_u);
        }

        // ---- epilogue (k_layer_conv's, the kind at run time: wave-uniform branches)
        auto epilogue = [&](auto KC) {
            constexpr int KIND = decltype(KC)::value;
            constexpr int NC = 32;
            // (the head partials are parked in the first chunk buffer: ROWS x 3 x 32 floats = 96 KiB (48 KiB) sits below the ring)
            __attribute__((address_space(3))) float *scratch = (__attribute__((address_space(3))) float *)lds;
#pragma unroll
            for (int g = 0; g < CT / 2; g++)
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
                    unsigned char *dst = outp + ((size_t)((obase >> 5) + g) * G::ROWS + board * 64 + 16 * pt + r) * G::GROW + q * 16;
                    f32x4v o[2] = {acc[pt][2 * g], acc[pt][2 * g + 1]};
                    if constexpr (KIND >= 2) {
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
                    for (int c = 0; c < NC; c++) v += scratch[i * NC + c];
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
        };
        if (kind == 0) epilogue(std::integral_constant<int, 0>{});
        else if (kind == 1) epilogue(std::integral_constant<int, 1>{});
        else if (kind == 2) epilogue(std::integral_constant<int, 2>{});
        else epilogue(std::integral_constant<int, 3>{});
        // this workgroup's stores have landed before any of its waves requests the next layer's first chunk
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

}  // namespace crl_tower

using namespace crl_tower;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float frnd(float s) { return ((int)(rnd() % 2001) - 1000) * 1e-3f * s; }

typedef void (*conv_t)(const unsigned char *, const unsigned char *, const float *, unsigned char *, const int *, const float *, const float *, float *, float *);
typedef void (*tower_t)(unsigned char *, unsigned char *, const unsigned char *, const float *, int, const int *, const float *, const float *, float *, int);

template <int NB>
static void run(int boards, int blocks, int reps)
{
    typedef LayerGeoT<NB> G;
    const int n_wg = boards / NB, n_layers = 1 + 2 * blocks;
    const size_t img_bytes = (size_t)n_wg * G::ACT_WG_BYTES;
    const size_t w_bytes = G::conv_bytes(4) + (size_t)2 * blocks * G::conv_bytes(8);
    // the input image (what k_layer_expand leaves: 4 chunks of 0/1 planes, lo = 0) -- random 0/1 here
    std::vector<_Float16> in(img_bytes / 2, (_Float16)0.f);
    for (size_t wg = 0; wg < (size_t)n_wg; wg++)
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < G::ROWS; row++)
                for (int e = 0; e < 32; e++)
                    in[wg * (G::ACT_WG_BYTES / 2) + ((size_t)c * G::ROWS + row) * 64 + e] = (_Float16)((rnd() & 7) == 0 ? 1.f : 0.f);
    // weight planes: random hi, small lo (the layout is whatever the kernels read: both paths read the same bytes)
    std::vector<_Float16> w(w_bytes / 2);
    const float s_stem = 1.0f / sqrtf(9.0f * 16), s_blk = 1.0f / sqrtf(9.0f * 256 * 0.5f);
    size_t at = 0;
    for (int L = 0; L < n_layers; L++) {
        const int chunks = L ? 8 : 4;
        for (int ct = 0; ct < chunks * 9; ct++)
            for (int part = 0; part < 2; part++)
                for (int i = 0; i < G::TILE / 2; i++)
                    w[at++] = (_Float16)(part ? frnd((L ? s_blk : s_stem) * 4e-4f) : frnd(L ? s_blk : s_stem));
    }
    std::vector<float> bias((size_t)n_layers * 256), head_w(3 * 256), head_b(3);
    for (auto &b : bias) b = frnd(0.05f);
    for (auto &v : head_w) v = frnd(0.1f);
    for (auto &v : head_b) v = frnd(0.1f);

    unsigned char *d_a, *d_b, *d_in, *d_w; float *d_bias, *d_hw, *d_hb, *d_ho1, *d_ho2;
    CK(hipMalloc(&d_a, img_bytes)); CK(hipMalloc(&d_b, img_bytes)); CK(hipMalloc(&d_in, img_bytes));
    CK(hipMemcpy(d_in, in.data(), img_bytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_w, w_bytes)); CK(hipMemcpy(d_w, w.data(), w_bytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_bias, bias.size() * 4)); CK(hipMemcpy(d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hw, 3072)); CK(hipMemcpy(d_hw, head_w.data(), 3072, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hb, 12)); CK(hipMemcpy(d_hb, head_b.data(), 12, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ho1, (size_t)boards * 192 * 4)); CK(hipMalloc(&d_ho2, (size_t)boards * 192 * 4));
    CK(hipMemset(d_ho1, 0xff, (size_t)boards * 192 * 4)); CK(hipMemset(d_ho2, 0xee, (size_t)boards * 192 * 4));

    const conv_t stem = (conv_t)k_layer_conv<4, 0, 0, NB>, c1 = (conv_t)k_layer_conv<8, 1, 0, NB>, c2 = (conv_t)k_layer_conv<8, 2, 0, NB>,
                 c3 = (conv_t)k_layer_conv<8, 3, 0, NB>;
    const tower_t tower = (tower_t)k_layer_tower<0, NB>;
    for (conv_t k : {stem, c1, c2, c3}) CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)tower, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));

    auto sequence = [&]() {          // as chessrl_amd/csrc/api.hip: layer_trunk_launch
        const unsigned char *lw = d_w;
        auto conv = [&](conv_t k, const unsigned char *i, unsigned char *o, int ci, bool last) {
            hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), G::LDS_BYTES, 0, i, lw, d_bias + (size_t)ci * 256, o, nullptr, d_hw, d_hb,
                               last ? d_ho1 : nullptr, nullptr);
        };
        conv(stem, d_b, d_a, 0, false);
        lw += G::conv_bytes(4);
        for (int blk = 0; blk < blocks; blk++) {
            conv(c1, d_a, d_b, 1 + 2 * blk, false);
            lw += G::conv_bytes(8);
            const bool last = blk + 1 == blocks;
            conv(last ? c3 : c2, d_b, d_a, 2 + 2 * blk, last);
            lw += G::conv_bytes(8);
        }
    };
    int stagger = 0;
    auto persistent = [&]() {
        hipLaunchKernelGGL(tower, dim3(n_wg), dim3(512), G::LDS_BYTES, 0, d_a, d_b, d_w, d_bias, blocks, nullptr, d_hw, d_hb, d_ho2, stagger);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best[2] = {1e9, 1e9};
    for (int round = 0; round < 3; round++)
        for (int which = 0; which < 2; which++) {
            CK(hipMemcpy(d_b, d_in, img_bytes, hipMemcpyDeviceToDevice));
            if (which) persistent(); else sequence();
            CK(hipDeviceSynchronize());
            CK(hipGetLastError());
            float total = 0;
            for (int i = 0; i < reps; i++) {
                CK(hipMemcpy(d_b, d_in, img_bytes, hipMemcpyDeviceToDevice));
                CK(hipEventRecord(e0));
                if (which) persistent(); else sequence();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                total += ms;
            }
            if (total / reps < best[which]) best[which] = total / reps;
        }
    CK(hipGetLastError());
    std::vector<float> h1((size_t)boards * 192), h2(h1.size());
    CK(hipMemcpy(h1.data(), d_ho1, h1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), d_ho2, h2.size() * 4, hipMemcpyDeviceToHost));
    size_t differ = 0, nonzero = 0; double mx = 0;
    for (size_t i = 0; i < h1.size(); i++) {
        differ += memcmp(&h1[i], &h2[i], 4) != 0;
        nonzero += h1[i] != 0.f;
        if (fabs(h1[i]) > mx) mx = fabs(h1[i]);
    }
    printf("%d blocks x 256, %d boards, %d boards per workgroup: launch sequence %.3f ms, one persistent launch %.3f ms (%+.1f %%); head outputs: "
           "%zu of %zu differ (%zu non-zero, max %.3g) -> %s\n", blocks, boards, NB, best[0], best[1], 100.0 * (best[1] / best[0] - 1.0),
           differ, h1.size(), nonzero, mx, differ || !std::isfinite(mx) || !nonzero ? "WRONG" : "bit-identical");
    // round 6: the same persistent launch with its first round of workgroups started in four phases
    for (int ticks : {500, 1000, 2000, 3500, 5000, 7000}) {
        stagger = ticks;
        double b = 1e9;
        for (int round = 0; round < 3; round++) {
            float total = 0;
            for (int i = 0; i < reps; i++) {
                CK(hipMemcpy(d_b, d_in, img_bytes, hipMemcpyDeviceToDevice));
                CK(hipEventRecord(e0));
                persistent();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                total += ms;
            }
            if (total / reps < b) b = total / reps;
        }
        CK(hipMemcpy(h2.data(), d_ho2, h2.size() * 4, hipMemcpyDeviceToHost));
        size_t d2 = 0;
        for (size_t i = 0; i < h1.size(); i++) d2 += memcmp(&h1[i], &h2[i], 4) != 0;
        printf("    persistent, first round staggered by (blockIdx & 3) x %d ticks of 10 ns: %.3f ms (%+.1f %% vs the launch sequence), %zu outputs differ\n",
               ticks, b, 100.0 * (b / best[0] - 1.0), d2);
        fflush(stdout);
    }
    hipFree(d_a); hipFree(d_b); hipFree(d_in); hipFree(d_w); hipFree(d_bias); hipFree(d_hw); hipFree(d_hb); hipFree(d_ho1); hipFree(d_ho2);
}

int main(int argc, char **argv)
{
    const int boards = argc > 1 ? atoi(argv[1]) : 4096, blocks = argc > 2 ? atoi(argv[2]) : 20, reps = argc > 3 ? atoi(argv[3]) : 5;
    run<4>(boards, blocks, reps);
    run<2>(boards >= 2048 ? 512 : boards, blocks, reps);
    return 0;
}
