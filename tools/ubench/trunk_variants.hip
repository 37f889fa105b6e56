// Scratch (GPU): time correct-result variants of the fused trunk kernels against production and
// check that their outputs are bit-identical.  hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=off -I chessrl_amd/csrc -I tools/ubench/r1_kernels tools/ubench/trunk_variants.hip -o tools/ubench/trunk_variants
//   ./trunk_variants [boards=4096] [reps=20]
#define CRL_HARNESS 1
#include "tower_pipe.hpp"
#include "tower_gen.hpp"
#include "tower_x16.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace crl_tower;
typedef void (*kern_t)(const unsigned char *, const unsigned char *, const float *, float *, int,
                       const float *, const float *, float *);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Bufs { unsigned char *planes, *wts, *wts16; float *bias, *head_w, *head_b, *head_out; };   // wts: round-1 tile order (32x32x16 kernels); wts16: plane order (x16)

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static double run(const char *name, kern_t k, int lds, int boards_per_wg, int F, int blocks, int boards, int reps,
                  const Bufs &b, std::vector<float> &out, const std::vector<float> *ref)
{
    const unsigned char *wsel = strstr(name, "x16") ? b.wts16 : b.wts;     // each kernel family reads its own image
    if (const char *only = getenv("ONLY")) {              // profiling: run just the kernels whose label matches
        if (!strstr(name, only)) return 0;
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; i++)
            hipLaunchKernelGGL(k, dim3(boards / boards_per_wg), dim3(512), lds, 0, b.planes, wsel, b.bias, nullptr, blocks, b.head_w, b.head_b, b.head_out);
        CK(hipDeviceSynchronize());
        printf("ran %s x3\n", name);
        return 0;
    }
    const char *match = getenv("MATCH");                  // A/B runs: time just the kernels whose label matches
    if (match && !strstr(name, match)) return 0;
    if (match && ref && ref->size() != (size_t)boards * 192) ref = nullptr;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(b.head_out, 0, (size_t)boards * 192 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9;
    for (int round = 0; round < 3; round++) {
        for (int i = 0; i < 2; i++)
            hipLaunchKernelGGL(k, dim3(boards / boards_per_wg), dim3(512), lds, 0, b.planes, wsel, b.bias, nullptr, blocks, b.head_w, b.head_b, b.head_out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++)
            hipLaunchKernelGGL(k, dim3(boards / boards_per_wg), dim3(512), lds, 0, b.planes, wsel, b.bias, nullptr, blocks, b.head_w, b.head_b, b.head_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    CK(hipGetLastError());
    out.resize((size_t)boards * 192);
    CK(hipMemcpy(out.data(), b.head_out, out.size() * 4, hipMemcpyDeviceToHost));
    const double flops = 2.0 * (73152.0 * F + 1152.0 * F * F * blocks + 192.0 * F) * boards;
    bool same = !ref || memcmp(ref->data(), out.data(), out.size() * 4) == 0;
    double sum = 0, maxd = 0, maxv = 0;
    for (size_t i = 0; i < out.size(); i++) {
        sum += out[i];
        if (ref) { double d = fabs((double)out[i] - (*ref)[i]); if (d > maxd) maxd = d; if (fabs((*ref)[i]) > maxv) maxv = fabs((*ref)[i]); }
    }
    char cmp[96];
    if (!ref) snprintf(cmp, sizeof cmp, "reference");
    else if (same) snprintf(cmp, sizeof cmp, "bit-identical");
    else snprintf(cmp, sizeof cmp, "max|d| %.3g of max %.3g", maxd, maxv);
    printf("%-34s %8.4f ms  %7.1f TFLOP/s  frac %.3f  %s (checksum %.6e)\n", name, best, flops / best / 1e9,
           flops / best / 1e9 / 2500.0, cmp, sum);
    fflush(stdout);
    return best;
}

static Bufs make(int F, int blocks, int boards)
{
    Bufs b;
    const int n_convs = 1 + 2 * blocks;
    const size_t wbytes = ((size_t)9 * 128 * F + (size_t)2 * blocks * 9 * F * F) * 2;
    std::vector<uint64_t> planes((size_t)boards * 128);
    for (auto &p : planes) { uint64_t v = 0; for (int i = 0; i < 64; i++) if (rnd() % 8 == 0) v |= 1ull << i; p = v; }
    std::vector<_Float16> w(wbytes / 2);
    const float scale = 1.5f / sqrtf(9.0f * F);
    for (auto &x : w) x = (_Float16)(((int)(rnd() % 2001) - 1000) * 1e-3f * scale);
    std::vector<float> bias((size_t)n_convs * F), hw(3 * F), hb(3);
    for (auto &x : bias) x = ((int)(rnd() % 201) - 100) * 1e-3f;
    for (auto &x : hw) x = ((int)(rnd() % 201) - 100) * 1e-3f;
    for (auto &x : hb) x = 0.1f;
    CK(hipMalloc(&b.planes, planes.size() * 8)); CK(hipMemcpy(b.planes, planes.data(), planes.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.wts, wbytes)); CK(hipMemcpy(b.wts, w.data(), wbytes, hipMemcpyHostToDevice));
    {   // the same weights as the x16 kernels' planes: [tile][sub-step][row][4 chunks][8], rows and chunks in LDS order
        const int KT = F == 256 ? 32 : 64, SPT = KT / 32;
        const size_t n_tiles = wbytes / 2 / ((size_t)F * KT);
        std::vector<_Float16> p(wbytes / 2);
        for (size_t t = 0; t < n_tiles; t++)
            for (int ks = 0; ks < SPT; ks++)
                for (int r = 0; r < F; r++) {
                    const int ch = Geo16<128, 4>::row_channel(r), sw = (0 - (r >> 2)) & 3;
                    for (int ph = 0; ph < 4; ph++)
                        for (int e = 0; e < 8; e++)
                            p[((t * SPT + ks) * F + r) * 32 + ph * 8 + e] =
                                w[(t * F + ch) * KT + ks * 32 + (ph ^ sw) * 8 + e];
                }
        CK(hipMalloc(&b.wts16, wbytes)); CK(hipMemcpy(b.wts16, p.data(), wbytes, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&b.bias, bias.size() * 4)); CK(hipMemcpy(b.bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.head_w, hw.size() * 4)); CK(hipMemcpy(b.head_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.head_b, 12)); CK(hipMemcpy(b.head_b, hb.data(), 12, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.head_out, (size_t)boards * 192 * 4));
    return b;
}

template <int F, int NB, int PAIR = 0, int GROUP = 0>
static void stamps(const Bufs &b, int blocks, int boards)
{
    if (getenv("MATCH")) return;
    typedef Geo16<F, NB> G;
    kern_t k = k_trunk_x16<F, NB, 1, 2, PAIR, GROUP>;
    const int lds = G::lds_bytes(GROUP ? (NB == 2 ? 12 : 9) : (PAIR ? 5 : 4));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int nwg = boards / NB;
    unsigned long long *dbg; CK(hipMalloc(&dbg, (size_t)nwg * 8 * 6 * 8));
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL(k, dim3(nwg), dim3(512), lds, 0, b.planes, b.wts16, b.bias, (float *)dbg, blocks, b.head_w, b.head_b, b.head_out);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)nwg * 48);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double loop = 0, epi = 0, tot = 0, ba = 0, wr = 0, vm = 0, sb = 0;
    for (int i = 0; i < nwg * 8; i++) {
        loop += h[i * 6]; epi += h[i * 6 + 1]; tot += h[i * 6 + 3]; ba += (double)(h[i * 6 + 2] >> 32); wr += (double)(h[i * 6 + 2] & 0xffffffffull);
        vm += h[i * 6 + 4]; sb += h[i * 6 + 5];
    }
    const int n = nwg * 8, convs = 1 + 2 * blocks;
    const double mf = (F == 64 ? (4.0 + 2.0 * blocks * 2) : F == 128 ? 4.0 * convs : (4.0 + 2.0 * blocks * 8)) * 9 * (G::PT * G::CT) * 16 * 2;
    printf("stamps x16<%d,%d%s>: per wave: main loops %.0f cycles (MFMA-paced minimum at 2 waves/SIMD %.0f), epilogues %.0f (%.0f per conv), "
           "whole kernel %.0f; loop share %.3f epilogue share %.3f\n", F, NB, PAIR ? ", pair sync" : (GROUP ? ", groups of 3" : ""), loop / n, mf, epi / n, epi / n / convs, tot / n,
           loop / tot, epi / tot);
    printf("   epilogue phases per conv: wait at the first barrier %.0f, convert + LDS writes %.0f, second barrier %.0f cycles\n",
           ba / n / convs, wr / n / convs, (epi - ba - wr) / n / convs);
    printf("   inside the loops: waiting for the weight DMA (vmcnt) %.0f cycles, at the tile barrier %.0f (incl. 2 stamp reads per sync)\n", vm / n, sb / n);
    CK(hipFree(dbg));
}

int main(int argc, char **argv)
{
    const int boards = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 20;
    std::vector<float> ref, out;
    {
        Bufs b = make(128, 10, boards);
        printf("== 10 x 128, %d boards\n", boards);
        run("k_trunk128_pipe<0,1> production", k_trunk128_pipe<0, 1>, P2_LDS_BYTES, 4, 128, 10, boards, reps, b, ref, nullptr);
        run("k_trunk128_pipe<0,1> again", k_trunk128_pipe<0, 1>, P2_LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("k_trunk_x16<128,4,1> 16x16x32", k_trunk_x16<128, 4, 1>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("k_trunk_x16<128,4,1,0,1> pair sync", k_trunk_x16<128, 4, 1, 0, 1>, Geo16<128, 4>::lds_bytes(5), 4, 128, 10, boards, reps, b, out, &ref);
        run("k_trunk_x16<128,4,1> again", k_trunk_x16<128, 4, 1>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("k_trunk_x16<128,4,1,1> alt issuer", k_trunk_x16<128, 4, 1, 1>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("k_trunk_x16<128,4,1> again", k_trunk_x16<128, 4, 1>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        stamps<128, 4>(b, 10, boards);
        stamps<128, 4, 1>(b, 10, boards);
        run("  x16<128,4> no staging (timing)", k_trunk_x16<128, 4, 1, 3>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("  x16<128,4> no barrier (timing)", k_trunk_x16<128, 4, 1, 4>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("  x16<128,4> no fragment reads (timing)", k_trunk_x16<128, 4, 1, 6>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("  x16<128,4> neither (timing)", k_trunk_x16<128, 4, 1, 5>, Geo16<128, 4>::LDS_BYTES, 4, 128, 10, boards, reps, b, out, &ref);
        run("k_trunk_x16<128,2,1> 512 boards", k_trunk_x16<128, 2, 1>, Geo16<128, 2>::LDS_BYTES, 2, 128, 10, 512, reps, b, out, nullptr);
    }
    if (argc > 3 && atoi(argv[3]) == 0) return 0;
    {
        Bufs b = make(256, 20, boards);
        printf("== 20 x 256, %d boards\n", boards);
        typedef Geo<256, 2> G;
        run("k_trunk_gen<256,2,1> production", k_trunk_gen<256, 2, 1>, G::LDS_BYTES, 2, 256, 20, boards, 5, b, ref, nullptr);
        run("k_trunk_gen<256,2,1> again", k_trunk_gen<256, 2, 1>, G::LDS_BYTES, 2, 256, 20, boards, 5, b, out, &ref);
        run("k_trunk_x16<256,2,1> 16x16x32", k_trunk_x16<256, 2, 1>, Geo16<256, 2>::LDS_BYTES, 2, 256, 20, boards, 5, b, out, &ref);
        stamps<256, 2>(b, 20, boards);
        stamps<256, 2, 1>(b, 20, boards);
        run("  x16<256,2> no staging (timing)", k_trunk_x16<256, 2, 1, 3>, Geo16<256, 2>::LDS_BYTES, 2, 256, 20, boards, 5, b, out, &ref);
        run("  x16<256,2> no barrier (timing)", k_trunk_x16<256, 2, 1, 4>, Geo16<256, 2>::LDS_BYTES, 2, 256, 20, boards, 5, b, out, &ref);
        run("  x16<256,2> neither (timing)", k_trunk_x16<256, 2, 1, 5>, Geo16<256, 2>::LDS_BYTES, 2, 256, 20, boards, 5, b, out, &ref);
        run("k_trunk_x16<256,2,1,0,1> pair sync", k_trunk_x16<256, 2, 1, 0, 1>, Geo16<256, 2>::lds_bytes(5), 2, 256, 20, boards, 5, b, out, &ref);
        run("k_trunk_x16<256,2,1> again", k_trunk_x16<256, 2, 1>, Geo16<256, 2>::LDS_BYTES, 2, 256, 20, boards, 5, b, out, &ref);
        run("k_trunk_x16<256,2,1,1> alt issuer", k_trunk_x16<256, 2, 1, 1>, Geo16<256, 2>::LDS_BYTES, 2, 256, 20, boards, 5, b, out, &ref);
    }
    {
        Bufs b = make(64, 6, boards);
        printf("== 6 x 64, %d boards (4-board workgroups), then 512 boards (2-board workgroups)\n", boards);
        typedef Geo<64, 4> G4; typedef Geo<64, 2> G2;
        run("k_trunk_gen<64,4,1> production", k_trunk_gen<64, 4, 1>, G4::LDS_BYTES, 4, 64, 6, boards, reps, b, ref, nullptr);
        run("k_trunk_x16<64,4,1> 16x16x32", k_trunk_x16<64, 4, 1>, Geo16<64, 4>::LDS_BYTES, 4, 64, 6, boards, reps, b, out, &ref);
        run("k_trunk_x16<64,4,1,0,0,1> groups of 3", k_trunk_x16<64, 4, 1, 0, 0, 1>, Geo16<64, 4>::lds_bytes(9), 4, 64, 6, boards, reps, b, out, &ref);
        stamps<64, 4>(b, 6, boards);
        stamps<64, 2>(b, 6, 512);
        stamps<64, 4, 0, 1>(b, 6, boards);
        stamps<64, 2, 0, 1>(b, 6, 512);
        run("k_trunk_x16<64,4,1,1> alt issuer", k_trunk_x16<64, 4, 1, 1>, Geo16<64, 4>::LDS_BYTES, 4, 64, 6, boards, reps, b, out, &ref);
        run("k_trunk_gen<64,2,1> production 512", k_trunk_gen<64, 2, 1>, G2::LDS_BYTES, 2, 64, 6, 512, reps, b, ref, nullptr);
        run("k_trunk_x16<64,2,1> 16x16x32 512", k_trunk_x16<64, 2, 1>, Geo16<64, 2>::LDS_BYTES, 2, 64, 6, 512, reps, b, out, &ref);
        run("k_trunk_x16<64,2,1,0,0,1> groups of 3, 512", k_trunk_x16<64, 2, 1, 0, 0, 1>, Geo16<64, 2>::lds_bytes(12), 2, 64, 6, 512, reps, b, out, &ref);
    }
    {
        Bufs b = make(128, 10, boards);
        printf("== 10 x 128 through the template, %d boards\n", boards);
        typedef Geo<128, 4> G;
        run("k_trunk_gen<128,4,1>", k_trunk_gen<128, 4, 1>, G::LDS_BYTES, 4, 128, 10, boards, reps, b, ref, nullptr);
    }
    return 0;
}
