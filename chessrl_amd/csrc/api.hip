// api.hip -- host side of libchessrl_hip.so: context, HBM pools, kernel launches and the
// C-ABI declared in include/chessrl_hip.h.  gfx950 only; no torch types cross this boundary.
#include "../../include/chessrl_hip.h"
#include "search.hpp"
#include "tower_x16.hpp"
#include "tower_layer.hpp"
#include "heads.hpp"
#include "train_ops.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

using namespace crl;

static_assert(sizeof(crl_board) == sizeof(Board), "crl_board layout");
static_assert(CRL_MAX_MOVES == MAX_MOVES && CRL_N_LABELS == N_LABELS && CRL_PLANES == PLANES,
              "header constants");

static thread_local std::string g_create_error;

// sha256 of the sources this library was compiled from (chessrl_amd/_lib.py passes it at build time; the
// marker makes it findable in the file without loading it)
#ifndef CRL_SOURCE_HASH
#define CRL_SOURCE_HASH "unstamped"
#endif
static const char g_source_hash[] = "CRL_SRC_HASH=" CRL_SOURCE_HASH;

struct crl_ctx {
    int device = 0;
    int W = 0;                    // slots in the active window [d.g0, d.g0 + W)
    hipStream_t stream = nullptr;
    Dev d{};
    std::vector<void *> allocs;
    std::string error;
    // device staging for host-facing calls
    u16 *t_moves = nullptr;       // [G][256]
    u16 *t_moves2 = nullptr;      // [G][256]
    int32_t *t_i32a = nullptr;    // [G][256]
    int32_t *t_i32b = nullptr;    // [G]
    int32_t *t_i32c = nullptr;    // [G]
    double *t_f64 = nullptr;      // [G][256]
    float *t_f32 = nullptr;       // [G][256]
    uint8_t *t_u8 = nullptr;      // [G]
    u16 *t_u16a = nullptr;        // [G]
    u16 *t_u16b = nullptr;        // [G]
    Board *t_boards = nullptr;    // [G]
};

// ---- netencoder.get_uci_labels (netencoder.py:94-134) as move ids ------------------------------
// Order is the contract: per from-square (file-major), destinations along the rank, the file,
// the a1-h8 diagonal, the h1-a8 diagonal, then 8 knight jumps; then per file the under-/promotion
// labels q,r,b,n x {straight, capture towards a, capture towards h} x {rank 2->1, rank 7->8}.
static void build_labels(std::vector<u16> &labels)
{
    labels.clear();
    static const int kn[8][2] = { { -2, -1 }, { -1, -2 }, { -2, 1 }, { 1, -2 },
                                  { 2, -1 }, { -1, 2 }, { 2, 1 }, { 1, 2 } };
    for (int f = 0; f < 8; f++)
        for (int r = 0; r < 8; r++) {
            std::vector<std::pair<int, int>> dst;
            for (int t = 0; t < 8; t++) dst.push_back({ t, r });
            for (int t = 0; t < 8; t++) dst.push_back({ f, t });
            for (int t = -7; t < 8; t++) dst.push_back({ f + t, r + t });
            for (int t = -7; t < 8; t++) dst.push_back({ f + t, r - t });
            for (auto &k : kn) dst.push_back({ f + k[0], r + k[1] });
            for (auto &q : dst) {
                if (q.first == f && q.second == r) continue;
                if (q.first < 0 || q.first > 7 || q.second < 0 || q.second > 7) continue;
                labels.push_back((u16)((r * 8 + f) | ((q.second * 8 + q.first) << 6)));
            }
        }
    static const int promo_code[4] = { 5, 4, 3, 2 };       // q r b n
    for (int f = 0; f < 8; f++)
        for (int pi = 0; pi < 4; pi++) {
            const int pc = promo_code[pi] << 12;
            labels.push_back((u16)((8 + f) | ((0 + f) << 6) | pc));
            labels.push_back((u16)((48 + f) | ((56 + f) << 6) | pc));
            if (f > 0) {
                labels.push_back((u16)((8 + f) | ((0 + f - 1) << 6) | pc));
                labels.push_back((u16)((48 + f) | ((56 + f - 1) << 6) | pc));
            }
            if (f < 7) {
                labels.push_back((u16)((8 + f) | ((0 + f + 1) << 6) | pc));
                labels.push_back((u16)((48 + f) | ((56 + f + 1) << 6) | pc));
            }
        }
}

static int fail(crl_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return fail(ctx, CRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T>
static hipError_t dalloc(crl_ctx *ctx, T **p, size_t count, bool zero = true)
{
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
    ctx->allocs.push_back(q);
    if (zero) {
        e = hipMemset(q, 0, bytes);
        if (e != hipSuccess) return e;
    }
    *p = (T *)q;
    return hipSuccess;
}

static const char *dev_err_name(int c)
{
    switch (c) {
    case DERR_NODE_POOL: return "node pool overflow (more than max_sims+1 nodes in one tree)";
    case DERR_EDGE_POOL: return "edge pool overflow";
    case DERR_PLY_POOL: return "game longer than max_plies";
    case DERR_STATE: return "search call out of order (reply/backup/advance state machine)";
    case DERR_BRANCH: return "more than 256 legal moves";
    case DERR_LABEL: return "legal move without a policy label";
    default: return "unknown device error";
    }
}

static int check_dev_error(crl_ctx *ctx)
{
    int32_t e = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&e, ctx->d.err, sizeof e, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (e != 0) return fail(ctx, e == DERR_STATE ? CRL_ERR_STATE : CRL_ERR_CAPACITY, dev_err_name(e));
    return CRL_OK;
}

#define LAUNCH(ctx, kern, ...)                                                          \
    do {                                                                                \
        hipLaunchKernelGGL(kern, dim3((ctx)->W), dim3(64), 0, (ctx)->stream, __VA_ARGS__); \
        HIP_TRY(ctx, hipGetLastError());                                                \
    } while (0)

extern "C" {

int crl_create(crl_ctx **out, int device, int max_games, int max_sims, int max_plies, uint32_t flags)
{
    if (!out || max_games < 1 || max_sims < 1 || max_sims > 32000 || max_plies < 2)
        return fail(nullptr, CRL_ERR_ARG, "crl_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, CRL_ERR_HIP, "crl_create: no HIP device visible (the HIP path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, CRL_ERR_ARG, "crl_create: bad device index");
    crl_ctx *ctx = new crl_ctx();
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e)); }
    Dev &d = ctx->d;
    d.G = max_games;
    d.N = max_sims + 1;
    d.ECAP = d.N * MAX_BRANCH;
    d.MAXPLY = max_plies;
    d.flags = flags;
    d.g0 = 0;
    d.plane_fmt = CRL_PLANES_F16;
    d.policy_fmt = CRL_POLICY_FULL;
    d.stats_s1 = d.stats_s2 = nullptr;
    ctx->W = max_games;
    const size_t G = d.G, GN = G * d.N, GE = G * (size_t)d.ECAP;
    bool ok = true;
#define A(ptr, count) ok = ok && (dalloc(ctx, &(ptr), (count)) == hipSuccess)
#define AN(ptr, count) ok = ok && (dalloc(ctx, &(ptr), (count), false) == hipSuccess)
    A(d.game, G); A(d.cur, G); A(d.hist, G * HIST_RING); A(d.hist_hash, G * HIST_RING);
    A(d.rec_moves, G * (size_t)d.MAXPLY);
    AN(d.node, GN);
    AN(d.edge, GE);
    AN(d.path_edge, GN); AN(d.path_node, GN);
    A(d.s1_moves, G * MAX_MOVES);
    A(d.lab_s1, G * MAX_MOVES); A(d.lab_s2, G * MAX_MOVES); A(d.lab_n1, G); A(d.lab_n2, G);
    A(d.counters, G * CNT_N); A(d.err, 1);
    A(ctx->t_moves, G * MAX_MOVES); A(ctx->t_moves2, G * MAX_MOVES); A(ctx->t_i32a, G * MAX_MOVES);
    A(ctx->t_i32b, G); A(ctx->t_i32c, G); A(ctx->t_f64, G * MAX_MOVES); A(ctx->t_f32, G * MAX_MOVES);
    A(ctx->t_u8, G); A(ctx->t_u16a, G); A(ctx->t_u16b, G); A(ctx->t_boards, G);
    u16 *lut = nullptr;
    A(lut, 5 * 4096);
#undef A
#undef AN
    if (!ok) {
        std::string msg = "crl_create: hipMalloc failed (pools need about " +
                          std::to_string((GE * sizeof(Edge) + GN * (sizeof(NodeRow) + 8)) >> 20) + " MiB)";
        for (void *p : ctx->allocs) (void)hipFree(p);
        delete ctx;
        return fail(nullptr, CRL_ERR_HIP, msg);
    }
    std::vector<u16> labels;
    build_labels(labels);
    std::vector<u16> h(5 * 4096, 0xFFFF);
    for (size_t i = 0; i < labels.size(); i++) {
        u32 mv = labels[i], p = (mv >> 12) & 7, slot = p ? 6 - p : 0;
        h[slot * 4096 + (mv & 4095)] = (u16)i;
    }
    e = hipMemcpy(lut, h.data(), h.size() * sizeof(u16), hipMemcpyHostToDevice);
    d.lut = lut;
    if (e != hipSuccess || labels.size() != N_LABELS) {
        for (void *p : ctx->allocs) (void)hipFree(p);
        delete ctx;
        return fail(nullptr, CRL_ERR_HIP, "crl_create: label table upload failed");
    }
    *out = ctx;
    int rc = crl_reset_games(ctx, nullptr);
    if (rc == CRL_OK) rc = crl_sync(ctx);
    if (rc != CRL_OK) { g_create_error = ctx->error; crl_destroy(ctx); *out = nullptr; }
    return rc;
}

void crl_destroy(crl_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (void *p : ctx->allocs) (void)hipFree(p);
    delete ctx;
}

int crl_set_stream(crl_ctx *ctx, void *hip_stream)
{
    if (!ctx) return CRL_ERR_ARG;
    ctx->stream = (hipStream_t)hip_stream;
    return CRL_OK;
}

int crl_sync(crl_ctx *ctx)
{
    if (!ctx) return CRL_ERR_ARG;
    return check_dev_error(ctx);
}

int crl_set_window(crl_ctx *ctx, int first, int count)
{
    if (!ctx || first < 0 || count < 1 || first + count > ctx->d.G)
        return fail(ctx, CRL_ERR_ARG, "crl_set_window: bad slot range");
    ctx->d.g0 = first;
    ctx->W = count;
    return CRL_OK;
}

int crl_set_plane_format(crl_ctx *ctx, int format)
{
    if (!ctx || (format != CRL_PLANES_F16 && format != CRL_PLANES_BITS))
        return fail(ctx, CRL_ERR_ARG, "crl_set_plane_format: bad argument");
    ctx->d.plane_fmt = format;
    return CRL_OK;
}

int crl_set_policy_format(crl_ctx *ctx, int format)
{
    if (!ctx || (format != CRL_POLICY_FULL && format != CRL_POLICY_LEGAL && format != CRL_POLICY_LEGAL_RAW))
        return fail(ctx, CRL_ERR_ARG, "crl_set_policy_format: bad argument");
    if (format == CRL_POLICY_LEGAL_RAW && (!ctx->d.stats_s1 || !ctx->d.stats_s2))
        return fail(ctx, CRL_ERR_STATE, "crl_set_policy_format: CRL_POLICY_LEGAL_RAW needs crl_set_policy_stats for both evaluations first");
    ctx->d.policy_fmt = format;
    return CRL_OK;
}

int crl_set_policy_stats(crl_ctx *ctx, int which, const void *dev_stats_f32)
{
    if (!ctx || (which != 0 && which != 1) || !dev_stats_f32)
        return fail(ctx, CRL_ERR_ARG, "crl_set_policy_stats: bad argument");
    (which ? ctx->d.stats_s2 : ctx->d.stats_s1) = (const float2 *)dev_stats_f32;
    return CRL_OK;
}

int crl_eval_labels(crl_ctx *ctx, int which, const uint16_t **dev_labels, const int32_t **dev_counts)
{
    if (!ctx || (which != 0 && which != 1) || !dev_labels || !dev_counts)
        return fail(ctx, CRL_ERR_ARG, "crl_eval_labels: bad argument");
    *dev_labels = which ? ctx->d.lab_s2 : ctx->d.lab_s1;
    *dev_counts = which ? ctx->d.lab_n2 : ctx->d.lab_n1;
    return CRL_OK;
}

int crl_copy_game(crl_ctx *ctx, int dst, int src)
{
    if (!ctx || dst < 0 || src < 0 || dst >= ctx->d.G || src >= ctx->d.G)
        return fail(ctx, CRL_ERR_ARG, "crl_copy_game: bad slot");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (dst != src) {
        hipLaunchKernelGGL(k_copy_game, dim3(1), dim3(64), 0, ctx->stream, ctx->d, dst, src);
        HIP_TRY(ctx, hipGetLastError());
    }
    return CRL_OK;
}

int crl_copy_game_from(crl_ctx *ctx, int dst, crl_ctx *src_ctx, int src)
{
    if (!ctx || !src_ctx || dst < 0 || src < 0 || dst >= ctx->d.G || src >= src_ctx->d.G)
        return fail(ctx, CRL_ERR_ARG, "crl_copy_game_from: bad slot");
    if (ctx->device != src_ctx->device)
        return fail(ctx, CRL_ERR_ARG, "crl_copy_game_from: both contexts must live on one GPU");
    if (src_ctx == ctx) return crl_copy_game(ctx, dst, src);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (src_ctx->stream != ctx->stream) HIP_TRY(ctx, hipStreamSynchronize(src_ctx->stream));
    hipLaunchKernelGGL(k_copy_game_across, dim3(1), dim3(64), 0, ctx->stream, ctx->d, dst, src_ctx->d, src);
    HIP_TRY(ctx, hipGetLastError());
    return check_dev_error(ctx);
}

const char *crl_last_error(crl_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }
const char *crl_source_hash(void) { return g_source_hash + 13; }
int crl_abi_version(void) { return CRL_ABI_VERSION; }
int crl_max_games(crl_ctx *ctx) { return ctx ? ctx->d.G : CRL_ERR_ARG; }
int crl_max_sims(crl_ctx *ctx) { return ctx ? ctx->d.N - 1 : CRL_ERR_ARG; }

int crl_uci_label_moves(uint16_t *moves_out)
{
    if (!moves_out) return CRL_ERR_ARG;
    std::vector<u16> labels;
    build_labels(labels);
    if (labels.size() != N_LABELS) return CRL_ERR_STATE;
    memcpy(moves_out, labels.data(), N_LABELS * sizeof(u16));
    return CRL_OK;
}

// ---- Game seam ----------------------------------------------------------------------------
int crl_reset_games(crl_ctx *ctx, const uint8_t *mask)
{
    if (!ctx) return CRL_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint8_t *dm = nullptr;
    if (mask) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->t_u8, mask, ctx->W, hipMemcpyHostToDevice, ctx->stream));
        dm = ctx->t_u8;
    }
    LAUNCH(ctx, k_set_positions, ctx->d, (const Board *)nullptr, dm, ctx->W, 1);
    if (mask) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host mask may be freed by caller
    return CRL_OK;
}

int crl_set_positions(crl_ctx *ctx, const crl_board *boards, int n)
{
    if (!ctx || !boards || n < 0 || n > ctx->W) return fail(ctx, CRL_ERR_ARG, "crl_set_positions: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->t_boards, boards, (size_t)n * sizeof(Board), hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, k_set_positions, ctx->d, (const Board *)ctx->t_boards, (const uint8_t *)nullptr, n, 0);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CRL_OK;
}

int crl_get_positions(crl_ctx *ctx, crl_board *boards_out, int n)
{
    if (!ctx || !boards_out || n < 0 || n > ctx->W) return fail(ctx, CRL_ERR_ARG, "crl_get_positions: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(boards_out, ctx->d.cur + ctx->d.g0, (size_t)n * sizeof(Board), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CRL_OK;
}

int crl_legal_moves(crl_ctx *ctx, uint16_t *moves, int32_t *counts)
{
    if (!ctx || !counts) return fail(ctx, CRL_ERR_ARG, "crl_legal_moves: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W;
    LAUNCH(ctx, k_legal_moves, ctx->d, ctx->t_moves, ctx->t_i32b);
    if (moves)
        HIP_TRY(ctx, hipMemcpyAsync(moves, ctx->t_moves, G * MAX_MOVES * sizeof(u16), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(counts, ctx->t_i32b, G * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return check_dev_error(ctx);
}

int crl_push_moves(crl_ctx *ctx, const uint16_t *moves, uint8_t *ok)
{
    if (!ctx || !moves || !ok) return fail(ctx, CRL_ERR_ARG, "crl_push_moves: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->t_u16a, moves, G * sizeof(u16), hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, k_push, ctx->d, (const u16 *)ctx->t_u16a, ctx->t_u8);
    HIP_TRY(ctx, hipMemcpyAsync(ok, ctx->t_u8, G, hipMemcpyDeviceToHost, ctx->stream));
    return check_dev_error(ctx);
}

int crl_push_sequences(crl_ctx *ctx, const uint16_t *moves, const int32_t *counts, int stride,
                       int32_t *pushed)
{
    if (!ctx || !moves || !counts || !pushed || stride < 1)
        return fail(ctx, CRL_ERR_ARG, "crl_push_sequences: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W;
    u16 *dseq = nullptr;                      // setup-time call: a scratch buffer per call is fine
    HIP_TRY(ctx, hipMalloc((void **)&dseq, G * (size_t)stride * sizeof(u16)));
    hipError_t e = hipMemcpyAsync(dseq, moves, G * (size_t)stride * sizeof(u16), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(ctx->t_i32b, counts, G * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_push_seq, dim3((unsigned)G), dim3(64), 0, ctx->stream, ctx->d, (const u16 *)dseq,
                           (const int32_t *)ctx->t_i32b, stride, ctx->t_i32c);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(pushed, ctx->t_i32c, G * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    hipError_t es = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dseq);
    if (e != hipSuccess) return fail(ctx, CRL_ERR_HIP, hipGetErrorString(e));
    if (es != hipSuccess) return fail(ctx, CRL_ERR_HIP, hipGetErrorString(es));
    return check_dev_error(ctx);
}

int crl_results(crl_ctx *ctx, int8_t *result)
{
    if (!ctx || !result) return fail(ctx, CRL_ERR_ARG, "crl_results: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LAUNCH(ctx, k_game_scalars, ctx->d, ctx->t_i32b, (int8_t *)ctx->t_u8);
    HIP_TRY(ctx, hipMemcpyAsync(result, ctx->t_u8, ctx->W, hipMemcpyDeviceToHost, ctx->stream));
    return check_dev_error(ctx);
}

int crl_records(crl_ctx *ctx, uint16_t *moves, int32_t *plies, int8_t *result)
{
    if (!ctx || !plies) return fail(ctx, CRL_ERR_ARG, "crl_records: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W;
    if (moves)
        HIP_TRY(ctx, hipMemcpyAsync(moves, ctx->d.rec_moves + (size_t)ctx->d.g0 * ctx->d.MAXPLY, G * ctx->d.MAXPLY * sizeof(u16), hipMemcpyDeviceToHost, ctx->stream));
    LAUNCH(ctx, k_game_scalars, ctx->d, ctx->t_i32b, (int8_t *)ctx->t_u8);
    HIP_TRY(ctx, hipMemcpyAsync(plies, ctx->t_i32b, G * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (result)
        HIP_TRY(ctx, hipMemcpyAsync(result, ctx->t_u8, G, hipMemcpyDeviceToHost, ctx->stream));
    return check_dev_error(ctx);
}

int crl_encode(crl_ctx *ctx, void *dev_planes_f16)
{
    if (!ctx || !dev_planes_f16) return fail(ctx, CRL_ERR_ARG, "crl_encode: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LAUNCH(ctx, k_encode_cur, ctx->d, dev_planes_f16);
    return CRL_OK;
}

int crl_greedy_moves(crl_ctx *ctx, const void *dev_policy_f32, const uint8_t *mask, int push, uint16_t *moves_out)
{
    if (!ctx || !dev_policy_f32) return fail(ctx, CRL_ERR_ARG, "crl_greedy_moves: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint8_t *dm = nullptr;
    if (mask) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->t_u8, mask, ctx->W, hipMemcpyHostToDevice, ctx->stream));
        dm = ctx->t_u8;
    }
    LAUNCH(ctx, k_greedy, ctx->d, (const float *)dev_policy_f32, dm, push, ctx->t_u16a);
    if (moves_out)
        HIP_TRY(ctx, hipMemcpyAsync(moves_out, ctx->t_u16a, (size_t)ctx->W * sizeof(u16), hipMemcpyDeviceToHost, ctx->stream));
    if (mask || moves_out) return check_dev_error(ctx);
    return CRL_OK;
}

// ---- SelfPlayTree seam ----------------------------------------------------------------------
int crl_search_begin(crl_ctx *ctx, void *dev_planes_f16)
{
    if (!ctx || !dev_planes_f16) return fail(ctx, CRL_ERR_ARG, "crl_search_begin: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LAUNCH(ctx, k_search_begin, ctx->d, dev_planes_f16);
    return CRL_OK;
}

int crl_search_root_priors(crl_ctx *ctx, const void *dev_policy_f32)
{
    if (!ctx || !dev_policy_f32) return fail(ctx, CRL_ERR_ARG, "crl_search_root_priors: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LAUNCH(ctx, k_root_priors, ctx->d, (const float *)dev_policy_f32);
    return CRL_OK;
}

int crl_sim_select_expand(crl_ctx *ctx, const void *dev_policy_s2_f32, const void *dev_value_s2_f32,
                          void *dev_planes_s1_f16)
{
    if (!ctx || !dev_planes_s1_f16) return fail(ctx, CRL_ERR_ARG, "crl_sim_select_expand: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LAUNCH(ctx, k_select_expand, ctx->d, (const float *)dev_policy_s2_f32, (const float *)dev_value_s2_f32,
           dev_planes_s1_f16);
    return CRL_OK;
}

int crl_sim_reply(crl_ctx *ctx, const void *dev_policy_s1_f32, void *dev_planes_s2_f16)
{
    if (!ctx || !dev_policy_s1_f32 || !dev_planes_s2_f16) return fail(ctx, CRL_ERR_ARG, "crl_sim_reply: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LAUNCH(ctx, k_reply, ctx->d, (const float *)dev_policy_s1_f32, dev_planes_s2_f16);
    return CRL_OK;
}

int crl_sim_backup(crl_ctx *ctx, const void *dev_policy_s2_f32, const void *dev_value_s2_f32)
{
    if (!ctx || !dev_policy_s2_f32 || !dev_value_s2_f32) return fail(ctx, CRL_ERR_ARG, "crl_sim_backup: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LAUNCH(ctx, k_backup, ctx->d, (const float *)dev_policy_s2_f32, (const float *)dev_value_s2_f32);
    return CRL_OK;
}

int crl_root_children(crl_ctx *ctx, int32_t *nchild, int32_t *visits, double *values, float *priors,
                      uint16_t *moves, uint16_t *replies, int32_t *root_visits)
{
    if (!ctx) return CRL_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W, R = G * MAX_MOVES;
    LAUNCH(ctx, k_root_children, ctx->d, ctx->t_i32b, ctx->t_i32a, ctx->t_f64, ctx->t_f32, ctx->t_moves,
           ctx->t_moves2, ctx->t_i32c);
    hipStream_t s = ctx->stream;
    if (nchild) HIP_TRY(ctx, hipMemcpyAsync(nchild, ctx->t_i32b, G * 4, hipMemcpyDeviceToHost, s));
    if (visits) HIP_TRY(ctx, hipMemcpyAsync(visits, ctx->t_i32a, R * 4, hipMemcpyDeviceToHost, s));
    if (values) HIP_TRY(ctx, hipMemcpyAsync(values, ctx->t_f64, R * 8, hipMemcpyDeviceToHost, s));
    if (priors) HIP_TRY(ctx, hipMemcpyAsync(priors, ctx->t_f32, R * 4, hipMemcpyDeviceToHost, s));
    if (moves) HIP_TRY(ctx, hipMemcpyAsync(moves, ctx->t_moves, R * 2, hipMemcpyDeviceToHost, s));
    if (replies) HIP_TRY(ctx, hipMemcpyAsync(replies, ctx->t_moves2, R * 2, hipMemcpyDeviceToHost, s));
    if (root_visits) HIP_TRY(ctx, hipMemcpyAsync(root_visits, ctx->t_i32c, G * 4, hipMemcpyDeviceToHost, s));
    return check_dev_error(ctx);
}

int crl_advance(crl_ctx *ctx, const int32_t *chosen, uint16_t *bm, uint16_t *am)
{
    if (!ctx || !chosen) return fail(ctx, CRL_ERR_ARG, "crl_advance: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->t_i32b, chosen, G * 4, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, k_advance, ctx->d, (const int32_t *)ctx->t_i32b, ctx->t_u16a, ctx->t_u16b);
    if (bm) HIP_TRY(ctx, hipMemcpyAsync(bm, ctx->t_u16a, G * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (am) HIP_TRY(ctx, hipMemcpyAsync(am, ctx->t_u16b, G * 2, hipMemcpyDeviceToHost, ctx->stream));
    return check_dev_error(ctx);
}

// ---- the move boundary in two synchronising calls (SelfPlayRunner.end_move) --------------------------------
int crl_end_move_fetch(crl_ctx *ctx, const void *dev_policy_s2_f32, const void *dev_value_s2_f32, int32_t *nchild,
                       int32_t *visits, int32_t *root_visits, int32_t *plies)
{
    if (!ctx || !dev_policy_s2_f32 || !dev_value_s2_f32 || !nchild || !visits || !root_visits || !plies)
        return fail(ctx, CRL_ERR_ARG, "crl_end_move_fetch: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W, R = G * MAX_MOVES;
    hipStream_t s = ctx->stream;
    LAUNCH(ctx, k_backup, ctx->d, (const float *)dev_policy_s2_f32, (const float *)dev_value_s2_f32);
    LAUNCH(ctx, k_root_children, ctx->d, ctx->t_i32b, ctx->t_i32a, ctx->t_f64, ctx->t_f32, ctx->t_moves,
           ctx->t_moves2, ctx->t_i32c);
    HIP_TRY(ctx, hipMemcpyAsync(nchild, ctx->t_i32b, G * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(visits, ctx->t_i32a, R * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(root_visits, ctx->t_i32c, G * 4, hipMemcpyDeviceToHost, s));
    // (the copies above are queued behind k_root_children; the staging arrays are free again for the scalars)
    LAUNCH(ctx, k_game_scalars, ctx->d, ctx->t_i32b, (int8_t *)ctx->t_u8);
    HIP_TRY(ctx, hipMemcpyAsync(plies, ctx->t_i32b, G * 4, hipMemcpyDeviceToHost, s));
    return check_dev_error(ctx);
}

int crl_advance_fetch(crl_ctx *ctx, const int32_t *chosen, uint16_t *bm, uint16_t *am, int8_t *results,
                      int32_t *legal_counts)
{
    if (!ctx || !chosen || !results || !legal_counts) return fail(ctx, CRL_ERR_ARG, "crl_advance_fetch: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = ctx->W;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->t_i32b, chosen, G * 4, hipMemcpyHostToDevice, s));
    LAUNCH(ctx, k_advance, ctx->d, (const int32_t *)ctx->t_i32b, ctx->t_u16a, ctx->t_u16b);
    if (bm) HIP_TRY(ctx, hipMemcpyAsync(bm, ctx->t_u16a, G * 2, hipMemcpyDeviceToHost, s));
    if (am) HIP_TRY(ctx, hipMemcpyAsync(am, ctx->t_u16b, G * 2, hipMemcpyDeviceToHost, s));
    LAUNCH(ctx, k_game_scalars, ctx->d, ctx->t_i32c, (int8_t *)ctx->t_u8);
    HIP_TRY(ctx, hipMemcpyAsync(results, ctx->t_u8, G, hipMemcpyDeviceToHost, s));
    LAUNCH(ctx, k_legal_moves, ctx->d, ctx->t_moves, ctx->t_i32b);
    HIP_TRY(ctx, hipMemcpyAsync(legal_counts, ctx->t_i32b, G * 4, hipMemcpyDeviceToHost, s));
    return check_dev_error(ctx);
}

int crl_counters(crl_ctx *ctx, uint64_t *out6)
{
    if (!ctx || !out6) return CRL_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<unsigned long long> h((size_t)ctx->d.G * CNT_N);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d.counters, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < CNT_N; k++) out6[k] = 0;
    for (int g = 0; g < ctx->d.G; g++)
        for (int k = 0; k < CNT_N; k++) out6[k] += h[(size_t)g * CNT_N + k];
    return CRL_OK;
}

// ---- tower seam (model.py) -------------------------------------------------------------------
int crl_trunk_forward(void *hip_stream, int filters, const void *dev_planes_f16,
                      const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                      int n_boards, int n_blocks, const void *dev_head_w_f32,
                      const void *dev_head_b_f32, void *dev_head_out_f32);

// small batches run half-size workgroups (see below); crl_trunk_set_small_batch turns that off
static std::atomic<int> g_small_batch{1};

int crl_trunk_set_small_batch(int enabled)
{
    g_small_batch.store(enabled ? 1 : 0);
    return CRL_OK;
}

// opt in to > 64 KiB of dynamic LDS once per (device, kernel)
static hipError_t allow_big_lds(const void *kern, int lds_bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<int, const void *>> ready;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &k : ready)
        if (k.first == dev && k.second == kern) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) ready.push_back({ dev, kern });
    return e;
}

// the dispatch rule of the fused trunk, in one place (trunk_forward launches by it and
// crl_trunk_kernel_name reports it)
static bool trunk_small_batch(int filters, int n_boards)
{
    return n_boards <= 128 * (filters == 256 ? 2 : 4) && g_small_batch.load() != 0;
}

struct TrunkPick { int nb, pair, group; };

static TrunkPick trunk_pick(int filters, int n_boards, bool split)
{
    const bool small = trunk_small_batch(filters, n_boards);
    if (split) {
        // split precision ("f16x3"): activation rows hold hi and lo, so 128 / 256 filters keep half the
        // boards per workgroup resident; 256 filters: plain ring of four (a fifth slot needs the zero rows
        // shortened to fit and was measured: 32.76 vs 32.82 ms, the kernel sits on the L2 -> LDS stream)
        if (filters == 256) return { 1, 0, 0 };
        if (filters == 128) return { 2, 1, 0 };
        return { small ? 2 : 4, 0, 0 };
    }
    // production: the 16x16x32-MFMA kernels of tower_x16.hpp for every filter count; 128 and 256
    // filters with one barrier per two weight tiles over a five-slot ring (PAIR)
    // 64 filters at four boards per workgroup: taps in groups of three over a nine-slot ring (GROUP);
    // at two boards per workgroup (batches <= 512) the plain ring of four measured faster
    // A batch that gives at most half of the 256 CUs a workgroup runs the half-size geometry
    // (half the boards per workgroup, twice the workgroups): C2's 512 boards are 128 workgroups of 4.
    const int nb = filters == 256 ? (small ? 1 : 2) : (small ? 2 : 4);
    return { nb, filters == 64 ? 0 : 1, (filters == 64 && nb == 4) ? 1 : 0 };
}

// The layer-wise split-precision trunk of 256 filters (csrc/tower_layer.hpp): one launch that expands the input planes into
// the first activation image, then one launch per convolution, ping-ponging between the two halves of the caller's workspace
// (crl_trunk_workspace_bytes); the second convolution of a block rewrites the block's input in place.
static bool trunk_is_layerwise(int filters, int flags)
{
    return filters == 256 && (flags & CRL_TRUNK_SPLIT);
}

extern "C++" {
// NB boards per workgroup; IDX: 0 = the whole batch, 2 / 3 = the boards of the list when it is short / long (tower_layer.hpp)
template <int NB, int IDX>
static int layer_trunk_launch(hipStream_t st, bool bits, const void *planes, const void *wts, const void *bias, void *out_f32,
                              int n_boards, int n_blocks, const void *head_w, const void *head_b, void *head_out,
                              void *workspace, const int32_t *list)
{
    typedef crl_tower::LayerGeoT<NB> G;
    // a short list is at most IDX_SMALL_MAX boards: that many workgroups cover it (and an idle launch of them costs least)
    const int n_wg = IDX == 2 ? std::min(n_boards / NB, crl_tower::IDX_SMALL_MAX / NB) : n_boards / NB;
    // the two activation images: 64 KiB per board each, whatever the geometry
    unsigned char *A = (unsigned char *)workspace, *B = A + (size_t)n_boards * (G::ACT_WG_BYTES / NB);
    typedef void (*conv_t)(const unsigned char *, const unsigned char *, const float *, unsigned char *, const int *,
                           const float *, const float *, float *, float *);
    typedef void (*expand_t)(const unsigned char *, unsigned char *, const int *);
    const expand_t ex = bits ? (expand_t)crl_tower::k_layer_expand<1, IDX, NB> : (expand_t)crl_tower::k_layer_expand<0, IDX, NB>;
    const conv_t stem = (conv_t)crl_tower::k_layer_conv<4, 0, IDX, NB>;
    const conv_t c1 = (conv_t)crl_tower::k_layer_conv<8, 1, IDX, NB>;
    const conv_t c2 = (conv_t)crl_tower::k_layer_conv<8, 2, IDX, NB>;
    conv_t c3 = (conv_t)crl_tower::k_layer_conv<8, 3, IDX, NB>;
    if constexpr (IDX == 0) {
        if (out_f32) c3 = (conv_t)crl_tower::k_layer_conv<8, 4, 0, NB>;       // (tests: the fp32 trunk as well)
    }
    for (conv_t k : { stem, c1, c2, c3 }) {
        hipError_t ea = allow_big_lds((const void *)k, G::LDS_BYTES);
        if (ea != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(ea));
    }
    const unsigned char *w = (const unsigned char *)wts;
    const float *b = (const float *)bias;
    const int *lst = (const int *)list;
    hipLaunchKernelGGL(ex, dim3(n_wg), dim3(512), 0, st, (const unsigned char *)planes, B, lst);
    auto conv = [&](conv_t k, const unsigned char *in, unsigned char *outp, int conv_index, bool last) {
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), G::LDS_BYTES, st, in, w, b + (size_t)conv_index * G::F, outp, lst,
                           (const float *)head_w, (const float *)head_b, last ? (float *)head_out : nullptr,
                           last ? (float *)out_f32 : nullptr);
    };
    conv(stem, B, A, 0, false);
    w += G::conv_bytes(4);
    for (int blk = 0; blk < n_blocks; blk++) {
        conv(c1, A, B, 1 + 2 * blk, false);
        w += G::conv_bytes(8);
        const bool last = blk + 1 == n_blocks;
        conv(last ? c3 : c2, B, A, 2 + 2 * blk, last);
        w += G::conv_bytes(8);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    return CRL_OK;
}
}  // extern "C++"

static int layer_trunk_forward(hipStream_t st, bool bits, const void *planes, const void *wts, const void *bias, void *out_f32,
                               int n_boards, int n_blocks, const void *head_w, const void *head_b, void *head_out,
                               void *workspace, size_t workspace_bytes, const int32_t *list)
{
    if (!workspace || n_blocks < 1 || !head_out)
        return fail(nullptr, CRL_ERR_ARG, "crl_trunk_forward: the layer-wise 256-filter split-precision trunk needs its "
                                           "workspace (crl_trunk_workspace_bytes), at least one residual block and head_out");
    // two activation images of 64 KiB per board, whatever the boards per workgroup (n_boards % 4 == 0 was checked)
    if (workspace_bytes < (size_t)2 * n_boards * (crl_tower::LayerGeo::ACT_WG_BYTES / 4))
        return fail(nullptr, CRL_ERR_ARG, "crl_trunk_forward: workspace_bytes is below crl_trunk_workspace_bytes(256, n_boards, "
                                           "CRL_TRUNK_SPLIT): the activation images of this batch would overrun the buffer");
    // Four boards per workgroup for full batches, two for batches of at most 512 boards.  The indexed launch of the hybrid mode
    // (a list is a few hundred boards of 4096) enqueues BOTH of its geometries, one board per workgroup for a list of at most
    // 256 boards and two beyond; the list, on the device, decides which sequence works (tower_layer.hpp: IDX 2 / 3).
    // Every output is accumulated in the same order in all three geometries: the same bits.
    if (list) {
        // (the planes of the whole tower through the memory-side cache first: tower_layer.hpp, k_layer_touch)
        hipLaunchKernelGGL(crl_tower::k_layer_touch, dim3(1024), dim3(256), 0, st, (const unsigned char *)wts,
                           crl_tower::LayerGeo::conv_bytes(4) + (size_t)2 * n_blocks * crl_tower::LayerGeo::conv_bytes(8), (const int *)list);
        const int rc = layer_trunk_launch<1, 2>(st, bits, planes, wts, bias, nullptr, n_boards, n_blocks, head_w, head_b, head_out,
                                                workspace, list);
        if (rc != CRL_OK) return rc;
        return layer_trunk_launch<2, 3>(st, bits, planes, wts, bias, nullptr, n_boards, n_blocks, head_w, head_b, head_out, workspace, list);
    }
    if (n_boards <= 512)
        return layer_trunk_launch<2, 0>(st, bits, planes, wts, bias, out_f32, n_boards, n_blocks, head_w, head_b, head_out, workspace, list);
    return layer_trunk_launch<4, 0>(st, bits, planes, wts, bias, out_f32, n_boards, n_blocks, head_w, head_b, head_out, workspace, list);
}

static int trunk_forward(void *hip_stream, int filters, const void *dev_planes_f16, int flags,
                         const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                         int n_boards, int n_blocks, const void *dev_head_w_f32,
                         const void *dev_head_b_f32, void *dev_head_out_f32, const int32_t *dev_index = nullptr,
                         void *dev_workspace = nullptr, size_t workspace_bytes = 0)
{
    if (filters != 64 && filters != 128 && filters != 256)
        return fail(nullptr, CRL_ERR_ARG, "crl_trunk_forward: the fused trunk covers 64, 128 and 256 filters");
    if (!dev_planes_f16 || !dev_wtiles_f16 || !dev_bias_f32 || (!dev_out_f32 && !dev_head_out_f32) ||
        (dev_head_out_f32 && (!dev_head_w_f32 || !dev_head_b_f32)) || n_boards < 4 ||
        n_boards % crl_tower::BOARDS_PER_WG != 0 || n_blocks < 0 ||
        1 + 2 * n_blocks > crl_tower::MAX_CONVS || (flags & ~(CRL_TRUNK_BITPLANES | CRL_TRUNK_SPLIT)) ||
        (dev_index && (dev_out_f32 || !dev_head_out_f32 || flags != (CRL_TRUNK_BITPLANES | CRL_TRUNK_SPLIT))))
        return fail(nullptr, CRL_ERR_ARG, "crl_trunk_forward: bad argument");
    typedef void (*kern_t)(const unsigned char *, const unsigned char *, const float *, float *, int,
                           const float *, const float *, float *);
    kern_t kern = nullptr;
    int lds_bytes = 0;
    const bool bits = flags & CRL_TRUNK_BITPLANES, split = flags & CRL_TRUNK_SPLIT;
    if (trunk_is_layerwise(filters, flags))
        return layer_trunk_forward((hipStream_t)hip_stream, bits, dev_planes_f16, dev_wtiles_f16, dev_bias_f32, dev_out_f32,
                                   n_boards, n_blocks, dev_head_w_f32, dev_head_b_f32, dev_head_out_f32, dev_workspace,
                                   workspace_bytes, dev_index);
    const TrunkPick pk = trunk_pick(filters, n_boards, split);
#define CRL_X16(F_, NB_, PAIR_, GROUP_, SPLIT_)                                                  \
    if (filters == F_ && pk.nb == NB_ && pk.pair == PAIR_ && pk.group == GROUP_ && split == (SPLIT_ != 0)) { \
        kern = bits ? crl_tower::k_trunk_x16<F_, NB_, 1, 0, PAIR_, GROUP_, SPLIT_>               \
                    : crl_tower::k_trunk_x16<F_, NB_, 0, 0, PAIR_, GROUP_, SPLIT_>;              \
        lds_bytes = crl_tower::Geo16<F_, NB_, SPLIT_>::lds_bytes(GROUP_ ? 9 : (PAIR_ ? 5 : crl_tower::PIPE_RING)); \
    }
    CRL_X16(256, 1, 1, 0, 0) CRL_X16(256, 2, 1, 0, 0)
    CRL_X16(64, 2, 0, 0, 0) CRL_X16(64, 4, 0, 1, 0)
    CRL_X16(128, 2, 1, 0, 0) CRL_X16(128, 4, 1, 0, 0)
    CRL_X16(128, 2, 1, 0, 1) CRL_X16(64, 2, 0, 0, 1) CRL_X16(64, 4, 0, 0, 1)       // (256 filters: tower_layer.hpp)
#undef CRL_X16
    if (dev_index) {
        // the list form (hybrid precision: the boards the single-MFMA pass could not decide): the split kernels
        // with their boards taken from the list the kernel finds in its `out` slot
        kern = nullptr;
#define CRL_X16I(F_, NB_, PAIR_)                                                                   \
        if (filters == F_ && pk.nb == NB_) kern = crl_tower::k_trunk_x16<F_, NB_, 1, 0, PAIR_, 0, 1, 1>;
        CRL_X16I(128, 2, 1) CRL_X16I(64, 2, 0) CRL_X16I(64, 4, 0)
#undef CRL_X16I
        dev_out_f32 = const_cast<int32_t *>(dev_index);
    }
    if (!kern) return fail(nullptr, CRL_ERR_ARG, "crl_trunk_forward: no kernel for this shape");
    hipError_t ea = allow_big_lds((const void *)kern, lds_bytes);
    if (ea != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(ea));
    hipLaunchKernelGGL(kern, dim3(n_boards / pk.nb), dim3(512),
                       lds_bytes, (hipStream_t)hip_stream,
                       (const unsigned char *)dev_planes_f16, (const unsigned char *)dev_wtiles_f16,
                       (const float *)dev_bias_f32, (float *)dev_out_f32, n_blocks,
                       (const float *)dev_head_w_f32, (const float *)dev_head_b_f32,
                       (float *)dev_head_out_f32);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    return CRL_OK;
}

int crl_trunk_forward_indexed(void *hip_stream, int filters, const void *dev_bitplanes_u64,
                              const void *dev_wtiles_f16x3, const void *dev_bias_f32, int n_boards, int n_blocks,
                              const void *dev_head_w_f32, const void *dev_head_b_f32, void *dev_head_out_f32,
                              const int32_t *dev_list, void *dev_workspace, size_t workspace_bytes)
{
    if (!dev_list) return fail(nullptr, CRL_ERR_ARG, "crl_trunk_forward_indexed: bad argument");
    return trunk_forward(hip_stream, filters, dev_bitplanes_u64, CRL_TRUNK_BITPLANES | CRL_TRUNK_SPLIT, dev_wtiles_f16x3,
                         dev_bias_f32, nullptr, n_boards, n_blocks, dev_head_w_f32, dev_head_b_f32, dev_head_out_f32,
                         dev_list, dev_workspace, workspace_bytes);
}

size_t crl_trunk_workspace_bytes(int filters, int n_boards, int flags)
{
    if (!trunk_is_layerwise(filters, flags) || n_boards < 1) return 0;
    return (size_t)2 * ((n_boards + 3) / 4) * crl_tower::LayerGeo::ACT_WG_BYTES;
}

int crl_reply_margin(void *hip_stream, const void *dev_priors_f32, const int32_t *dev_counts, int n_boards,
                     const float *dev_log_margin_f32, int rows_are_logits, int32_t *dev_list)
{
    if (!dev_priors_f32 || !dev_counts || n_boards < 1 || !dev_list || !dev_log_margin_f32)
        return fail(nullptr, CRL_ERR_ARG, "crl_reply_margin: bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(crl_heads::k_zero_word, dim3(1), dim3(64), 0, st, (int *)dev_list);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    hipLaunchKernelGGL(crl_heads::k_reply_margin, dim3((unsigned)((n_boards + 3) / 4)), dim3(256), 0, st,
                       (const float *)dev_priors_f32, (const int *)dev_counts, n_boards, dev_log_margin_f32,
                       rows_are_logits ? 1 : 0, (int *)dev_list);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    return CRL_OK;
}

int crl_trunk_kernel_name(int filters, int n_boards, int flags, char *buf, int buf_len)
{
    if ((filters != 64 && filters != 128 && filters != 256) || n_boards < 4 || !buf || buf_len < 1 ||
        (flags & ~(CRL_TRUNK_BITPLANES | CRL_TRUNK_SPLIT)))
        return fail(nullptr, CRL_ERR_ARG, "crl_trunk_kernel_name: bad argument");
    if (trunk_is_layerwise(filters, flags)) {
        // one forward = k_layer_expand + the stem + 2 launches per block; the block convolutions dominate
        // (named as rocprofv3 prints them: <CHUNKS, KIND, IDX, NB>; NB as layer_trunk_forward picks it)
        const int nb = n_boards <= 512 ? 2 : 4;
        snprintf(buf, (size_t)buf_len, "k_layer_conv<8, 1|2|3, 0, %d> (+ k_layer_conv<4, 0, 0, %d>, k_layer_expand<%d, 0, %d>)",
                 nb, nb, (flags & CRL_TRUNK_BITPLANES) ? 1 : 0, nb);
        return CRL_OK;
    }
    const TrunkPick pk = trunk_pick(filters, n_boards, flags & CRL_TRUNK_SPLIT);
    snprintf(buf, (size_t)buf_len, "k_trunk_x16<%d, %d, %d, 0, %d, %d, %d, 0>", filters, pk.nb,
             (flags & CRL_TRUNK_BITPLANES) ? 1 : 0, pk.pair, pk.group, (flags & CRL_TRUNK_SPLIT) ? 1 : 0);
    return CRL_OK;
}

int crl_trunk_forward_x(void *hip_stream, int filters, int flags, const void *dev_planes,
                        const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                        int n_boards, int n_blocks, const void *dev_head_w_f32,
                        const void *dev_head_b_f32, void *dev_head_out_f32, void *dev_workspace, size_t workspace_bytes)
{
    return trunk_forward(hip_stream, filters, dev_planes, flags, dev_wtiles_f16, dev_bias_f32, dev_out_f32,
                         n_boards, n_blocks, dev_head_w_f32, dev_head_b_f32, dev_head_out_f32, nullptr, dev_workspace,
                         workspace_bytes);
}

int crl_trunk_forward(void *hip_stream, int filters, const void *dev_planes_f16,
                      const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                      int n_boards, int n_blocks, const void *dev_head_w_f32,
                      const void *dev_head_b_f32, void *dev_head_out_f32)
{
    return trunk_forward(hip_stream, filters, dev_planes_f16, 0, dev_wtiles_f16, dev_bias_f32, dev_out_f32,
                         n_boards, n_blocks, dev_head_w_f32, dev_head_b_f32, dev_head_out_f32);
}

int crl_trunk_forward_bitplanes(void *hip_stream, int filters, const void *dev_bitplanes_u64,
                                const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                                int n_boards, int n_blocks, const void *dev_head_w_f32,
                                const void *dev_head_b_f32, void *dev_head_out_f32)
{
    return trunk_forward(hip_stream, filters, dev_bitplanes_u64, CRL_TRUNK_BITPLANES, dev_wtiles_f16, dev_bias_f32, dev_out_f32,
                         n_boards, n_blocks, dev_head_w_f32, dev_head_b_f32, dev_head_out_f32);
}

// Batches of at most g_sliced_max boards run the heads as label slices x board blocks + a normalising
// pass (csrc/heads.hpp: k_heads_sliced, k_policy_normalise) when the caller hands over the scratch for
// the slice statistics; larger ones the one-pass kernels (every CU already has a workgroup there).
static std::atomic<int> g_sliced_max{2048};

int crl_heads_set_sliced_max(int boards)
{
    g_sliced_max.store(boards < 0 ? 0 : boards);
    return CRL_OK;
}

static bool heads_sliced(int n_boards) { return n_boards <= g_sliced_max.load(); }

// raw: (LEGAL, sliced path only) leave the LOGITS in the priors rows and the slice statistics in the scratch;
// the consumers normalise on read (CRL_POLICY_LEGAL_RAW) and the normalising pass is not launched
static int heads_forward(void *hip_stream, const void *act, int n_boards, const void *pol_wp, const void *pol_bias,
                         const void *val_w1p, const void *val_b1, const void *val_w2b2, const uint16_t *labels,
                         const int32_t *counts, void *pol_out, void *val_out, void *scratch, bool legal, bool raw,
                         const char *who)
{
    if (!act || n_boards < 1 || !pol_wp || !pol_bias || !pol_out || (legal && (!labels || !counts)) ||
        (val_out && (!val_w1p || !val_b1 || !val_w2b2)))
        return fail(nullptr, CRL_ERR_ARG, who);
    if (raw && (!scratch || !heads_sliced(n_boards)))
        return fail(nullptr, CRL_ERR_ARG, "crl_heads_forward_legal_raw: needs the scratch and a batch the sliced heads serve (crl_heads_raw_supported)");
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned blocks = (unsigned)((n_boards + 15) / 16);
    if (scratch && heads_sliced(n_boards)) {
        const dim3 grid(crl_heads::N_SLICES + (val_out ? 1 : 0), blocks);
        if (legal) {
            hipLaunchKernelGGL(crl_heads::k_heads_sliced<true>, grid, dim3(256), 0, st, (const float *)act, n_boards,
                               (const unsigned char *)pol_wp, (const float *)pol_bias, (float *)pol_out, (float2 *)scratch,
                               (const unsigned short *)labels, (const int *)counts, (const unsigned char *)val_w1p,
                               (const float *)val_b1, (const float *)val_w2b2, (float *)val_out);
            if (!raw)
                hipLaunchKernelGGL(crl_heads::k_policy_normalise<true>, dim3((n_boards + 3) / 4), dim3(256), 0, st,
                                   (float *)pol_out, (const float2 *)scratch, n_boards, (const int *)counts);
        } else {
            hipLaunchKernelGGL(crl_heads::k_heads_sliced<false>, grid, dim3(256), 0, st, (const float *)act, n_boards,
                               (const unsigned char *)pol_wp, (const float *)pol_bias, (float *)pol_out, (float2 *)scratch,
                               (const unsigned short *)nullptr, (const int *)nullptr, (const unsigned char *)val_w1p,
                               (const float *)val_b1, (const float *)val_w2b2, (float *)val_out);
            hipLaunchKernelGGL(crl_heads::k_policy_normalise<false>, dim3((n_boards + 3) / 4), dim3(256), 0, st,
                               (float *)pol_out, (const float2 *)scratch, n_boards, (const int *)nullptr);
        }
    } else {
        // one launch: the value head rides along as extra workgroups behind the policy's (8 board blocks each)
        const unsigned vblocks = val_out ? (blocks + 7) / 8 : 0;
        if (legal) {
            auto kern = crl_heads::k_policy_head<1, true>;
            hipError_t ea = allow_big_lds((const void *)kern, crl_heads::LEGAL_LDS_BYTES);
            if (ea != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(ea));
            hipLaunchKernelGGL(kern, dim3(blocks + vblocks), dim3(512), crl_heads::LEGAL_LDS_BYTES, st,
                               (const float *)act, n_boards, (const unsigned char *)pol_wp, (const float *)pol_bias,
                               (float *)pol_out, (const unsigned short *)labels, (const int *)counts, (int)blocks,
                               (const unsigned char *)val_w1p, (const float *)val_b1, (const float *)val_w2b2,
                               (float *)val_out);
        } else {
            hipLaunchKernelGGL(crl_heads::k_policy_head<1>, dim3(blocks + vblocks), dim3(512), 0, st,
                               (const float *)act, n_boards, (const unsigned char *)pol_wp, (const float *)pol_bias,
                               (float *)pol_out, (const unsigned short *)nullptr, (const int *)nullptr, (int)blocks,
                               (const unsigned char *)val_w1p, (const float *)val_b1, (const float *)val_w2b2,
                               (float *)val_out);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    return CRL_OK;
}

int crl_heads_raw_supported(int n_boards) { return n_boards >= 1 && heads_sliced(n_boards) ? 1 : 0; }

int crl_heads_forward(void *hip_stream, const void *dev_head_act_f32, int n_boards,
                      const void *dev_policy_wp_f16, const void *dev_policy_bias_f32,
                      const void *dev_value_w1p_f16, const void *dev_value_b1_f32,
                      const void *dev_value_w2b2_f32, void *dev_policy_out_f32, void *dev_value_out_f32,
                      void *dev_scratch_f32)
{
    return heads_forward(hip_stream, dev_head_act_f32, n_boards, dev_policy_wp_f16, dev_policy_bias_f32,
                         dev_value_w1p_f16, dev_value_b1_f32, dev_value_w2b2_f32, nullptr, nullptr,
                         dev_policy_out_f32, dev_value_out_f32, dev_scratch_f32, false, false, "crl_heads_forward: bad argument");
}

int crl_heads_forward_legal(void *hip_stream, const void *dev_head_act_f32, int n_boards,
                            const void *dev_policy_wp_f16, const void *dev_policy_bias_f32,
                            const void *dev_value_w1p_f16, const void *dev_value_b1_f32,
                            const void *dev_value_w2b2_f32, const uint16_t *dev_labels,
                            const int32_t *dev_counts, void *dev_priors_out_f32, void *dev_value_out_f32,
                            void *dev_scratch_f32)
{
    return heads_forward(hip_stream, dev_head_act_f32, n_boards, dev_policy_wp_f16, dev_policy_bias_f32,
                         dev_value_w1p_f16, dev_value_b1_f32, dev_value_w2b2_f32, dev_labels, dev_counts,
                         dev_priors_out_f32, dev_value_out_f32, dev_scratch_f32, true, false,
                         "crl_heads_forward_legal: bad argument");
}

int crl_heads_forward_legal_raw(void *hip_stream, const void *dev_head_act_f32, int n_boards,
                                const void *dev_policy_wp_f16, const void *dev_policy_bias_f32,
                                const void *dev_value_w1p_f16, const void *dev_value_b1_f32,
                                const void *dev_value_w2b2_f32, const uint16_t *dev_labels,
                                const int32_t *dev_counts, void *dev_logits_out_f32, void *dev_value_out_f32,
                                void *dev_stats_out_f32)
{
    return heads_forward(hip_stream, dev_head_act_f32, n_boards, dev_policy_wp_f16, dev_policy_bias_f32,
                         dev_value_w1p_f16, dev_value_b1_f32, dev_value_w2b2_f32, dev_labels, dev_counts,
                         dev_logits_out_f32, dev_value_out_f32, dev_stats_out_f32, true, true,
                         "crl_heads_forward_legal_raw: bad argument");
}

static int train_op(void *hip_stream, const void *src, void *dst, int n_boards, int channels, bool forward)
{
    if (!src || !dst || n_boards < 1 || channels < 4 || channels % 4)
        return fail(nullptr, CRL_ERR_ARG, "crl_im2col3x3_f32 / crl_col2im3x3_f32: bad argument");
    const int c4 = channels / 4;
    const long long total = (long long)n_boards * 64 * c4 * (forward ? 9 : 1);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (forward)
        hipLaunchKernelGGL(crl_train::k_im2col3x3, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream,
                           (const float4 *)src, (float4 *)dst, total, c4);
    else
        hipLaunchKernelGGL(crl_train::k_col2im3x3, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream,
                           (const float4 *)src, (float4 *)dst, total, c4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    return CRL_OK;
}

// ---- measurement: (id, device wall clock) appended to a ring, capturable into a hipGraph (bench.py) ---------------------
__global__ void k_stamp(unsigned long long *ring, unsigned capacity, unsigned id)
{
    const unsigned long long n = ring[0];
    ring[2 + 2 * (n % capacity)] = id;
    ring[3 + 2 * (n % capacity)] = wall_clock64();
    ring[0] = n + 1;
}

int crl_stamp(void *hip_stream, uint64_t *dev_ring, uint32_t capacity, uint32_t id)
{
    if (!dev_ring || capacity < 1) return fail(nullptr, CRL_ERR_ARG, "crl_stamp: bad argument");
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)hip_stream, (unsigned long long *)dev_ring, capacity, id);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    return CRL_OK;
}

int crl_stamp_clock_khz(int device)
{
    int khz = 0;
    hipError_t e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device);
    if (e != hipSuccess) return fail(nullptr, CRL_ERR_HIP, hipGetErrorString(e));
    return khz;
}

int crl_im2col3x3_f32(void *hip_stream, const void *dev_x_f32, void *dev_cols_f32, int n_boards, int channels)
{
    return train_op(hip_stream, dev_x_f32, dev_cols_f32, n_boards, channels, true);
}

int crl_col2im3x3_f32(void *hip_stream, const void *dev_gcols_f32, void *dev_gx_f32, int n_boards, int channels)
{
    return train_op(hip_stream, dev_gcols_f32, dev_gx_f32, n_boards, channels, false);
}

}  // extern "C"
