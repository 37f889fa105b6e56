// Scratch (GPU), round 3: the two bounded experiments on the production trunk kernels
// (k_trunk_x16 with pair publishing) at 10 x 128 and 20 x 256, 4096 boards:
//   ALT 7  staging only: weight DMA + barriers + epilogues, no fragment reads, no MFMAs (timing only)
//   ALT 8  round 2's schedule: weight-fragment reads of the next sub-step in one clump before the MFMAs of half 1;
//          production (ALT 0) now issues them one by one between those MFMAs (bit-identical)
// and the time of the split-precision (f16x3) kernels on a random weight image.
// (A third variant was measured with this harness and removed again: warming each XCD's L2 for the weight
// tiles 8 / 16 tiles ahead, 1/32 of the lines per workgroup through 4-byte LDS-DMAs: +1.0 ... +1.5 % SLOWER
// at both sizes -- profiles/r03/trunk_r3_l2_prefetch.log.  The L2 -> LDS stream is not waiting for the fabric.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I chessrl_amd/csrc tools/ubench/trunk_r3.hip -o tools/ubench/trunk_r3
//   ./trunk_r3 [boards=4096] [reps=20]
#define CRL_HARNESS 1
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

struct Bufs { unsigned char *planes, *wts; float *bias, *head_w, *head_b, *head_out; };
static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static Bufs make(int F, int blocks, int boards, int image_mult)
{
    Bufs b;
    const int n_convs = 1 + 2 * blocks;
    const size_t wbytes = ((size_t)9 * 128 * F + (size_t)2 * blocks * 9 * F * F) * 2 * image_mult;
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
    CK(hipMalloc(&b.bias, bias.size() * 4)); CK(hipMemcpy(b.bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.head_w, hw.size() * 4)); CK(hipMemcpy(b.head_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.head_b, 12)); CK(hipMemcpy(b.head_b, hb.data(), 12, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.head_out, (size_t)boards * 192 * 4));
    return b;
}

static double run(const char *name, kern_t k, int lds, int nb, int F, int blocks, int boards, int reps,
                  const Bufs &b, std::vector<float> &out, const std::vector<float> *ref)
{
    const char *match = getenv("MATCH");
    if (match && !strstr(name, match)) return 0;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(b.head_out, 0, (size_t)boards * 192 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9;
    for (int round = 0; round < 3; round++) {
        for (int i = 0; i < 2; i++)
            hipLaunchKernelGGL(k, dim3(boards / nb), dim3(512), lds, 0, b.planes, b.wts, b.bias, nullptr, blocks, b.head_w, b.head_b, b.head_out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++)
            hipLaunchKernelGGL(k, dim3(boards / nb), dim3(512), lds, 0, b.planes, b.wts, b.bias, nullptr, blocks, b.head_w, b.head_b, b.head_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    CK(hipGetLastError());
    out.resize((size_t)boards * 192);
    CK(hipMemcpy(out.data(), b.head_out, out.size() * 4, hipMemcpyDeviceToHost));
    const double flops = 2.0 * (73152.0 * F + 1152.0 * F * F * blocks + 192.0 * F) * boards;
    const bool same = ref && memcmp(ref->data(), out.data(), out.size() * 4) == 0;
    printf("%-44s %8.4f ms  %7.1f TFLOP/s (algorithmic)  frac %.3f  %s\n", name, best, flops / best / 1e9,
           flops / best / 1e9 / 2500.0, !ref ? "reference" : (same ? "bit-identical" : "DIFFERENT (timing-only build)"));
    fflush(stdout);
    return best;
}

int main(int argc, char **argv)
{
    const int boards = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 20;
    std::vector<float> ref, out;
    for (int rep = 0; rep < 2; rep++) {
        Bufs b = make(128, 10, boards, 1);
        printf("== 10 x 128, %d boards (pass %d)\n", boards, rep);
        const int lds = Geo16<128, 4>::lds_bytes(5);
        run("x16<128,4> pair (production)", k_trunk_x16<128, 4, 1, 0, 1>, lds, 4, 128, 10, boards, reps, b, ref, nullptr);
        run("x16<128,4> pair, ALT 8 clumped w reads (r2)", k_trunk_x16<128, 4, 1, 8, 1>, lds, 4, 128, 10, boards, reps, b, out, &ref);
        run("x16<128,4> pair (production) again", k_trunk_x16<128, 4, 1, 0, 1>, lds, 4, 128, 10, boards, reps, b, out, &ref);
        run("x16<128,4> pair, ALT 7 staging only", k_trunk_x16<128, 4, 1, 7, 1>, lds, 4, 128, 10, boards, reps, b, out, &ref);
    }
    {
        Bufs b = make(256, 20, boards, 1);
        printf("== 20 x 256, %d boards\n", boards);
        const int lds = Geo16<256, 2>::lds_bytes(5);
        run("x16<256,2> pair (production)", k_trunk_x16<256, 2, 1, 0, 1>, lds, 2, 256, 20, boards, 5, b, ref, nullptr);
        run("x16<256,2> pair, ALT 8 clumped w reads (r2)", k_trunk_x16<256, 2, 1, 8, 1>, lds, 2, 256, 20, boards, 5, b, out, &ref);
        run("x16<256,2> pair (production) again", k_trunk_x16<256, 2, 1, 0, 1>, lds, 2, 256, 20, boards, 5, b, out, &ref);
        run("x16<256,2> pair, ALT 7 staging only", k_trunk_x16<256, 2, 1, 7, 1>, lds, 2, 256, 20, boards, 5, b, out, &ref);
    }
    {   // split precision: three MFMAs per product over a 2x weight image (random values: time only)
        Bufs b = make(128, 10, boards, 2);
        printf("== f16x3 (split operands), %d boards\n", boards);
        run("x16<128,2> pair split, 10 x 128", k_trunk_x16<128, 2, 1, 0, 1, 0, 1>, Geo16<128, 2, 1>::lds_bytes(5), 2, 128, 10, boards, reps, b, ref, nullptr);
        Bufs c = make(256, 20, boards, 2);
        run("x16<256,1> split, 20 x 256", k_trunk_x16<256, 1, 1, 0, 0, 0, 1>, Geo16<256, 1, 1>::lds_bytes(4), 1, 256, 20, boards, 3, c, ref, nullptr);
        Bufs d = make(64, 6, boards, 2);
        run("x16<64,4> split, 6 x 64", k_trunk_x16<64, 4, 1, 0, 0, 0, 1>, Geo16<64, 4, 1>::lds_bytes(4), 4, 64, 6, boards, reps, d, ref, nullptr);
    }
    return 0;
}
