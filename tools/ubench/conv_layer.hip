// Scratch (GPU), round 5: harness of the layer-wise split-precision 256 -> 256 convolution (conv_layer.hpp) --
// VERDICT r4 #1's bounded experiment.  ONE conv at `boards` boards incl. activation read + write; checked against a
// CPU evaluation of sampled outputs; timed as: loop alone (MODE 0), conv1 of a block (MODE 1: ReLU, hi/lo written),
// conv2 (MODE 2: + fp32 skip read, ReLU, hi/lo + skip written).  Stop rule: build into the product only if a layer
// takes <= 0.62 ms at 4096 boards (25 ms per 41-conv tower; k_trunk_x16<256,1,SPLIT> = 33.7 ms).
// The kernel is the product's (chessrl_amd/csrc/tower_layer.hpp: k_layer_conv<8, 1 | 2, 0>).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I chessrl_amd/csrc tools/ubench/conv_layer.hip -o tools/ubench/conv_layer
//   ./conv_layer [boards=4096] [reps=20]
#define CRL_HARNESS 1
#include "tower_layer.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace crl_tower;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float frnd(float s) { return ((int)(rnd() % 2001) - 1000) * 1e-3f * s; }

typedef void (*kern_t)(const unsigned char *, const unsigned char *, const float *, unsigned char *, const int *, const float *, const float *, float *, float *);

int main(int argc, char **argv)
{
    typedef LayerGeo G;
    typedef Geo16<256, 1, 1> WG;
    const int boards = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int n_wg = boards / 4;
    const size_t act_halves = (size_t)n_wg * G::ACT_WG_BYTES / 2;
    // activations: value v (post-ReLU-like, >= 0, ~25 % zeros), stored as hi = fp16(v), lo = fp16(v - hi)
    std::vector<_Float16> act(act_halves);
    for (size_t wg = 0; wg < (size_t)n_wg; wg++)
        for (int c = 0; c < 8; c++)
            for (int row = 0; row < 256; row++)
                for (int e = 0; e < 32; e++) {
                    float v = frnd(1.0f);
                    v = v < -0.5f ? 0.f : fabsf(v) * 1.37f;
                    const _Float16 hi = (_Float16)v;
                    const size_t at = ((wg * 8 + c) * 256 + row) * 64;
                    act[at + e] = hi;
                    act[at + 32 + e] = (_Float16)(v - (float)hi);
                }
    // weights in image order [chunk][tap][part][row][phys 16-B chunk][8]
    std::vector<_Float16> wimg(G::conv_bytes(8) / 2);
    std::vector<float> W((size_t)256 * 256 * 9), Whi(W.size()), Wlo(W.size());      // [o][cin][tap]
    const float scale = 1.5f / sqrtf(9.0f * 256);
    for (auto &w : W) w = frnd(scale);
    for (size_t i = 0; i < W.size(); i++) { Whi[i] = (float)(_Float16)W[i]; Wlo[i] = (float)(_Float16)(W[i] - Whi[i]); }
    for (int c = 0; c < 8; c++)
        for (int t = 0; t < 9; t++)
            for (int part = 0; part < 2; part++)
                for (int row = 0; row < 256; row++)
                    for (int phys = 0; phys < 4; phys++)
                        for (int e = 0; e < 8; e++) {
                            const int o = WG::row_channel(row), logical = phys ^ ((0 - (row >> 2)) & 3);
                            const int cin = 32 * c + 8 * logical + e;
                            const float v = (part ? Wlo : Whi)[((size_t)o * 256 + cin) * 9 + t];
                            wimg[((((size_t)(c * 9 + t) * 2 + part) * 256 + row) * 4 + phys) * 8 + e] = (_Float16)v;
                        }
    std::vector<float> bias(256);
    for (auto &b : bias) b = frnd(0.1f);
    // the block's input X (KIND 2 reads it from the OUTPUT image and rewrites it in place): another image like `act`
    std::vector<_Float16> xin(act_halves);
    for (size_t i = 0; i < act_halves; i += 64)
        for (int e = 0; e < 32; e++) {
            const float v = fabsf(frnd(1.0f));
            xin[i + e] = (_Float16)v;
            xin[i + 32 + e] = (_Float16)(v - (float)xin[i + e]);
        }

    unsigned char *d_act, *d_out, *d_w; float *d_bias;
    CK(hipMalloc(&d_act, act_halves * 2)); CK(hipMemcpy(d_act, act.data(), act_halves * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, act_halves * 2)); CK(hipMemset(d_out, 0, act_halves * 2));
    CK(hipMalloc(&d_w, wimg.size() * 2)); CK(hipMemcpy(d_w, wimg.data(), wimg.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_bias, 1024)); CK(hipMemcpy(d_bias, bias.data(), 1024, hipMemcpyHostToDevice));

    const double flops_alg = 2.0 * 64 * 9 * 256.0 * 256.0 * boards, flops_issued = 3.0 * flops_alg;
    std::vector<_Float16> out(act_halves);
    for (int mode = 0; mode <= 2; mode++) {
        kern_t k = mode == 0 ? (kern_t)k_layer_conv<8, 0, 0> : (mode == 1 ? (kern_t)k_layer_conv<8, 1, 0> : (kern_t)k_layer_conv<8, 2, 0>);
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        double best = 1e9;
        for (int round = 0; round < 3; round++) {
            for (int i = 0; i < 2; i++)
                hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), G::LDS_BYTES, 0, d_act, d_w, d_bias, d_out, nullptr, nullptr, nullptr, nullptr, nullptr);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; i++)
                hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), G::LDS_BYTES, 0, d_act, d_w, d_bias, d_out, nullptr, nullptr, nullptr, nullptr, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / reps < best) best = ms / reps;
        }
        CK(hipGetLastError());
        printf("conv 256->256 split, %d boards, KIND %d (%s): %8.4f ms  issued %7.1f TFLOP/s = %.3f of 2.5 PF, algorithmic %.3f  [stop rule: <= 0.62 ms]\n",
               boards, mode, mode == 0 ? "linear, hi/lo out" : (mode == 1 ? "conv1: ReLU, hi/lo out" : "conv2: + X (hi + lo) in, ReLU, in place"),
               best, flops_issued / best / 1e9, flops_issued / best / 1e9 / 2500.0, flops_alg / best / 1e9 / 2500.0);
        fflush(stdout);
        // ---- check sampled outputs against a CPU evaluation (double accumulation of the same three products)
        CK(hipMemcpy(d_out, xin.data(), act_halves * 2, hipMemcpyHostToDevice));      // X (only KIND 2 reads it)
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), G::LDS_BYTES, 0, d_act, d_w, d_bias, d_out, nullptr, nullptr, nullptr, nullptr, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), d_out, act_halves * 2, hipMemcpyDeviceToHost));
        double worst = 0; int bad = 0;
        for (int s = 0; s < 3000; s++) {
            const int wg = s < 1000 ? (s & 1 ? n_wg - 1 : 0) : rnd() % n_wg;
            const int b = rnd() % 4, p = s < 64 ? s : rnd() % 64, o = rnd() % 256;
            double sum = bias[o];
            for (int t = 0; t < 9; t++) {
                const int dy = t / 3 - 1, dx = t % 3 - 1, py = (p >> 3) + dy, px = (p & 7) + dx;
                if (py < 0 || py > 7 || px < 0 || px > 7) continue;
                const int row = b * 64 + py * 8 + px;
                for (int cin = 0; cin < 256; cin++) {
                    const size_t at = (((size_t)wg * 8 + (cin >> 5)) * 256 + row) * 64 + (cin & 31);
                    const double hi = (float)act[at], lo = (float)act[at + 32];
                    const size_t wi = ((size_t)o * 256 + cin) * 9 + t;
                    sum += hi * Whi[wi] + lo * Whi[wi] + hi * Wlo[wi];
                }
            }
            const size_t at = (((size_t)wg * 8 + (o >> 5)) * 256 + b * 64 + p) * 64 + (o & 31);
            if (mode == 2) sum += (double)(float)xin[at] + (double)(float)xin[at + 32];
            const double want = mode == 0 ? sum : (sum > 0 ? sum : 0);
            const double got = (double)(float)out[at] + (double)(float)out[at + 32];
            const double err = fabs(got - want);
            if (err > worst) worst = err;
            if (err > 2e-5 * (1.0 + fabs(want))) bad++;
        }
        printf("    check: 3000 sampled outputs vs CPU: max |err| %.3g (hi + lo), %d beyond 2e-5 relative -> %s\n", worst, bad, bad ? "WRONG" : "ok");
        fflush(stdout);
    }
    {   // proxy for a 128 -> 128 layer-wise convolution at four boards per workgroup: the two-board geometry (wave tile 64 x 64,
        // six sub-steps of 8 MFMAs) over 4 K-chunks does the same work per wave and writes the same bytes per workgroup; 2048
        // workgroups instead of 1024, so HALF its time estimates that convolution (its weight planes would be half as large)
        typedef LayerGeoT<2> G2;
        for (int mode = 1; mode <= 2; mode++) {
            kern_t k = mode == 1 ? (kern_t)k_layer_conv<4, 1, 0, 2> : (kern_t)k_layer_conv<4, 2, 0, 2>;
            CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, G2::LDS_BYTES));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            double best = 1e9;
            for (int round = 0; round < 3; round++) {
                hipLaunchKernelGGL(k, dim3(boards / 2), dim3(512), G2::LDS_BYTES, 0, d_act, d_w, d_bias, d_out, nullptr, nullptr, nullptr, nullptr, nullptr);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; i++)
                    hipLaunchKernelGGL(k, dim3(boards / 2), dim3(512), G2::LDS_BYTES, 0, d_act, d_w, d_bias, d_out, nullptr, nullptr, nullptr, nullptr, nullptr);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / reps < best) best = ms / reps;
            }
            CK(hipGetLastError());
            printf("proxy 128-filter layer-wise conv (k_layer_conv<4, %d, 0, 2>, %d boards, 128 in x 256 out): %.4f ms -> a 128 -> 128 convolution ~ %.4f ms\n",
                   mode, boards, best, best / 2);
        }
    }
#if defined(CRL_LAYER_STAMPS)
    {   // in-kernel cycle stamps (hipcc ... -DCRL_LAYER_STAMPS): per wave [loop, epilogue, vmcnt wait, barrier, DMA requests]
        unsigned long long *d_dbg; const size_t n_dbg = (size_t)n_wg * 8 * 5;
        CK(hipMalloc(&d_dbg, n_dbg * 8));
        for (int mode = 1; mode <= 2; mode++) {
            kern_t k = mode == 1 ? (kern_t)k_layer_conv<8, 1, 0> : (kern_t)k_layer_conv<8, 2, 0>;
            CK(hipMemset(d_dbg, 0, n_dbg * 8));
            for (int i = 0; i < 3; i++)
                hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), G::LDS_BYTES, 0, d_act, d_w, d_bias, d_out, nullptr, nullptr, nullptr, nullptr, (float *)d_dbg);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> dbg(n_dbg);
            CK(hipMemcpy(dbg.data(), d_dbg, n_dbg * 8, hipMemcpyDeviceToHost));
            double sum[5] = {0, 0, 0, 0, 0};
            for (size_t i = 0; i < n_dbg; i++) sum[i % 5] += (double)dbg[i];
            const double nw = (double)n_wg * 8;
            printf("stamps KIND %d, mean cycles per wave: loop %.0f (vmcnt wait %.0f = %.1f %%, barrier %.0f = %.1f %%, DMA requests %.0f = %.1f %%; "
                   "72 taps x 96 MFMAs x 16 = 110592 MFMA-issue cycles = %.1f %% of the loop, x 2 waves per SIMD), epilogue %.0f = %.1f %% of loop + epilogue\n",
                   mode, sum[0] / nw, sum[2] / nw, 100 * sum[2] / sum[0], sum[3] / nw, 100 * sum[3] / sum[0], sum[4] / nw, 100 * sum[4] / sum[0],
                   100 * 110592.0 / (sum[0] / nw), sum[1] / nw, 100 * sum[1] / (sum[0] + sum[1]));
        }
    }
#endif
    return 0;
}
