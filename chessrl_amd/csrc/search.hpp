// search.hpp -- device kernels of the lockstep self-play simulation loop.
//
// One 64-lane wavefront (one 64-thread workgroup) per game.  Reference symbols
// (paths relative to /root/reference/src/chessrl/):
//
//   k_search_begin     Tree.__init__                          mctree.py:105-111
//   k_root_priors      _update_prior for the root             mctree.py:298-303
//   k_select_expand    backprop (pending) + select + expand   mctree.py:216-257,278-296
//   k_reply            opponent reply half of expand, Node()  mctree.py:244-250,28-37
//   k_backup           simulate + backprop + child priors     mctree.py:259-303
//   k_root_children    [c.visits for c in root.children]      mctree.py:178,313-315
//   k_advance          gam.move(bm); gam.move(am)             selfplay.py:77-78
//   k_encode_cur       netencoder.get_game_state              netencoder.py:72-91
//   k_legal/k_push/... Game.get_legal_moves/move/get_result   game.py:28-57,92-109
//   k_greedy           AgentDistributed.best_move(real_game)  agentdistributed.py:56-58
//
// Float contract of get_value (mctree.py:71-87), reproduced with explicit
// round-to-nearest intrinsics (and the file is built with -ffp-contract=off):
//   Q = value / (1 + visits)                       float64 divide
//   U = (10 * prior) * (sqrt(sum) / (1 + visits))  10*prior in float32 (numpy>=2) or
//                                                  float64 (numpy 1.x flag); rest float64
//   sum = visits of the child's own children = visits-1 (0 for a terminal child), which
//   is exact in sequential mode: every simulation through a non-terminal node after the
//   one that created it visits exactly one of its children.
#pragma once
#include "movegen.hpp"
#include "slices.hpp"
#include "state.hpp"

namespace crl {

struct WaveLds {
    u16 mv[MAX_MOVES];
    Board enc[9];
    u64 pl[PLANES];
};

__device__ inline int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// ---- where does the position `tp` tree-plies after the root live? -------------------------
// tp >= 1: node path_node[(tp+1)/2], S1 if tp is odd else S2.  tp <= 0: game history ring
// at ply ply0 + tp, where ply0 is the game ply of tree-ply 0.
struct PastRef { const Board *b; const u64 *h; bool valid; };

__device__ inline PastRef past_ref(const Dev &d, int g, int tp, int ply0, int ply_ring_top)
{
    PastRef r;
    if (tp >= 1) {
        int node = d.path_node[(size_t)g * d.N + ((tp + 1) >> 1)];
        size_t ni = (size_t)g * d.N + node;
        r.b = (tp & 1) ? &d.node[ni].s1 : &d.node[ni].s2;
        r.h = (tp & 1) ? &d.node[ni].h1 : &d.node[ni].h2;
        r.valid = true;
    } else {
        int p = ply0 + tp;
        r.valid = p >= 0 && p > ply_ring_top - HIST_RING && p <= ply_ring_top;
        size_t hi = (size_t)g * HIST_RING + (p & (HIST_RING - 1));
        r.b = d.hist + hi;
        r.h = d.hist_hash + hi;
    }
    return r;
}

// earlier occurrences of position b (python-chess is_repetition: same _transposition_key),
// looking back over the reversible plies only; lanes test distances 2,4,6,... in parallel
__device__ inline int count_prior(const Dev &d, int g, const Board &b, u64 h, int tp, int ply0,
                                  int ring_top, int lane)
{
    const int clock = (int)st_clock(b.state);
    int cnt = 0;
    if (clock < 8) return 0;                 // a 5th occurrence needs >= 8 reversible plies
    for (int base = 0; 2 * (base + 1) <= clock; base += 64) {
        const int dist = 2 * (base + lane + 1);
        bool hit = false;
        if (dist <= clock) {
            PastRef r = past_ref(d, g, tp - dist, ply0, ring_top);
            if (r.valid && *r.h == h) {
                Board o = *r.b;
                hit = same_key(o, b);
            }
        }
        cnt += popc(__ballot(hit));
    }
    return cnt;
}

struct PosEval { Board b; u64 hash; int n; int result; };

// legal moves (into s.mv), derived ep bit, hash, repetition and Game.get_result of `b`
__device__ inline PosEval eval_position(const Dev &d, int g, Board b, int tp, int ply0,
                                        int ring_top, int lane, WaveLds &s)
{
    MoveGenInfo mi = wave_movegen(b, lane, s.mv);
    b.state = (b.state & ~(1u << 20)) | ((mi.ep_legal ? 1u : 0u) << 20);
    PosEval e;
    e.hash = board_hash(b);
    int rep = 1 + count_prior(d, g, b, e.hash, tp, ply0, ring_top, lane);
    e.result = position_result(b, mi.n, mi.in_check, rep);
    e.n = mi.n;
    e.b = b;
    __syncthreads();                          // s.mv visible to every lane
    return e;
}

// ---- netencoder.get_game_state (netencoder.py:13-91) -> fp16 NHWC [8][8][128] ----------------
// Channels: 0-6 black {no black piece here, P,N,B,R,Q,K}, 7-13 white likewise, then the same
// 14 planes for each of the 8 previous positions (zeros where the move stack is shorter),
// 126 = side to move is white, 127 = zero pad.  Row 0 = rank 8: spatial index = sq ^ 56.
// Lane l always owns channel group l&15, so its 8 plane bitboards stay in registers; each of the
// 16 store iterations writes one contiguous 1 KiB per wave.
__device__ inline void encode_position(const Dev &d, int g, const Board &b, int tp, int ply0,
                                       int ring_top, int lane, WaveLds &s, void *planes_out, int row)
{
    bool valid = false;
    if (lane < 9) {
        Board e = b;
        valid = true;
        if (lane > 0) {
            PastRef r = past_ref(d, g, tp - lane, ply0, ring_top);
            valid = r.valid;
            if (valid) e = *r.b;
        }
        s.enc[lane] = e;
    }
    const u32 vmask = (u32)__ballot(valid);
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int c = lane + 64 * half;
        u64 v = 0;
        if (c < 126) {
            const int i = c / 14, k = c % 14, t = k % 7;
            const Board &e = s.enc[i];
            const u64 occ = occupied(e);
            const u64 own = k >= 7 ? e.white : (occ & ~e.white);
            if ((vmask >> i) & 1) v = t == 0 ? ~own : (e.bb[t - 1] & own);
        } else if (c == 126) {
            v = st_turn(b.state) ? ~0ull : 0ull;
        }
        s.pl[c] = v;
    }
    __syncthreads();
    if (d.plane_fmt) {
        // compact form for the fused trunk (crl_trunk_forward_bitplanes): the 128 plane bitboards,
        // 1 KiB per position instead of 16 KiB; the trunk kernel expands them into LDS itself
        u64 *bits = (u64 *)planes_out + (size_t)row * PLANES;
        bits[lane] = s.pl[lane];
        bits[lane + 64] = s.pl[lane + 64];
        __syncthreads();
        return;
    }
    const int cg = lane & 15;
    u64 p8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) p8[k] = s.pl[cg * 8 + k];
    // (not unrolled: sixteen loop-invariant 64-bit store addresses, hoisted to the top of
    // k_select_expand, were what spilled 28 VGPRs there)
    uint4 *out = (uint4 *)planes_out + (size_t)row * (64 * PLANES * 2 / 16);
#pragma unroll 1
    for (int t = 0; t < 16; t++) {
        const int sq = (t * 4 + (lane >> 4)) ^ 56;
        u32 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            w[k] = (((p8[2 * k] >> sq) & 1) ? 0x3C00u : 0u) |
                   (((p8[2 * k + 1] >> sq) & 1) ? 0x3C000000u : 0u);
        out[t * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
}

// ---- shared pieces ---------------------------------------------------------------------------
__device__ inline void dev_error(const Dev &d, int code)
{
    atomicCAS(d.err, 0, code);
}

// Descent hint kept in the spare word of an edge record.  get_best_child needs, of the node it descends
// into, the slot of its edge records and their number -- both in that node's own record, i.e. behind a
// second dependent HBM read per tree level.  Once a node is fully expanded (from then on select only ever
// scans it, mctree.py:216-231) those two numbers are frozen, so they are copied into the parent's edge
// record, which the scan of the level above has just read: one dependent read per level instead of two.
constexpr u32 HINT_FULL = 1u << 31;
constexpr int HINT_EDGE_BITS = 23;                 // edge slots per game < 2^23 (38 000 simulations per move)
__device__ inline u32 hint_pack(int edge0, int nmoves) { return HINT_FULL | ((u32)nmoves << HINT_EDGE_BITS) | (u32)edge0; }
static_assert(MAX_BRANCH < 256, "nmoves fits the hint");

__device__ inline void init_edges(const Dev &d, size_t eb, int edge0, int n, const u16 *mv, int lane)
{
    for (int j = lane; j < n; j += 64) {
        Edge e;
        e.value = 0.0; e.visits = 0;
        e.prior = 1.0f;                      // Node.prior = 1 (mctree.py:35)
        e.move = mv[j]; e.child = CHILD_NONE; e.pad = 0;
        d.edge[eb + edge0 + j] = e;
    }
}

// fmt = CRL_POLICY_FULL: pol is a full policy[row][1968], gathered at the labels of the node's moves;
// fmt = CRL_POLICY_LEGAL: pol is priors[row][MAX_MOVES], entry j for the node's legal move j (the evaluator
// gathered them from the label list the search kernel that created the position wrote);
// fmt = CRL_POLICY_LEGAL_RAW: the same rows holding LOGITS; `stats` are the board's slice statistics and
// the probability is formed here (csrc/slices.hpp: the arithmetic of the normalising pass, same bits)
constexpr int FMT_FULL = 0, FMT_LEGAL = 1, FMT_LEGAL_RAW = 2;

__device__ inline void gather_priors(const Dev &d, int row, size_t eb, int edge0, int n,
                                     const float *pol, int lane, int fmt, const float2 *stats = nullptr)
{
    if (fmt == FMT_LEGAL_RAW) {
        const crl_slices::Norm nm = crl_slices::norm_of(stats + (size_t)row * crl_slices::N_SLICES);
        for (int j = lane; j < n; j += 64)
            d.edge[eb + edge0 + j].prior = crl_slices::prob(pol[(size_t)row * MAX_MOVES + j], nm);
        return;
    }
    const bool legal = fmt == FMT_LEGAL;
    for (int j = lane; j < n; j += 64) {
        size_t e = eb + edge0 + j;
        if (legal) {
            d.edge[e].prior = pol[(size_t)row * MAX_MOVES + j];
            continue;
        }
        int lab = label_of(d, d.edge[e].move);
        if (lab >= N_LABELS) { dev_error(d, DERR_LABEL); lab = 0; }
        d.edge[e].prior = pol[(size_t)row * N_LABELS + lab];
    }
}

// labels of the n moves in mv -> lab[row][0..n), count[row] = n (the evaluator's gather list)
__device__ inline void write_labels(const Dev &d, int row, const u16 *mv, int n, u16 *lab, int32_t *count, int lane)
{
    for (int i = lane; i < n; i += 64) {
        int l = label_of(d, mv[i]);
        if (l >= N_LABELS) { dev_error(d, DERR_LABEL); l = 0; }
        lab[(size_t)row * MAX_MOVES + i] = (u16)l;
    }
    if (lane == 0) count[row] = n;
}

// index of the first maximum of policy[label(m)] over the n moves in mv (np.argmax)
__device__ inline int argmax_policy(const Dev &d, int row, const u16 *mv, int n, const float *pol,
                                    int lane, int fmt, const float2 *stats = nullptr)
{
    float best = -__builtin_inff();
    int bi = 0x7FFFFFFF;
    crl_slices::Norm nm = {0.f, 0.f};
    if (fmt == FMT_LEGAL_RAW) nm = crl_slices::norm_of(stats + (size_t)row * crl_slices::N_SLICES);
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        if (i < n) {
            float p;
            if (fmt == FMT_LEGAL_RAW) {
                // the argmax runs over the PROBABILITIES the normalising pass would have stored (two
                // logits may round to one probability; np.argmax then takes the first)
                p = crl_slices::prob(pol[(size_t)row * MAX_MOVES + i], nm);
            } else if (fmt == FMT_LEGAL) {
                p = pol[(size_t)row * MAX_MOVES + i];
            } else {
                int lab = label_of(d, mv[i]);
                if (lab >= N_LABELS) { dev_error(d, DERR_LABEL); lab = 0; }
                p = pol[(size_t)row * N_LABELS + lab];
            }
            if (p > best || bi == 0x7FFFFFFF) { best = p; bi = i; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        float ob = __shfl_xor(best, o);
        int oi = __shfl_xor(bi, o);
        bool take = oi != 0x7FFFFFFF && (bi == 0x7FFFFFFF || ob > best || (ob == best && oi < bi));
        if (take) { best = ob; bi = oi; }
    }
    return bi;
}

// push a legal move onto the game (python-chess Board.push + result bookkeeping)
__device__ inline void game_push(const Dev &d, int g, const Board &b, u32 mv, int lane, WaveLds &s)
{
    const int p = d.game[g].ply;
    Board nb = apply_move(b, mv);
    PosEval e = eval_position(d, g, nb, 0, p + 1, p, lane, s);
    if (lane == 0) {
        if (p + 1 > d.MAXPLY) { dev_error(d, DERR_PLY_POOL); }
        else {
            size_t hi = (size_t)g * HIST_RING + ((p + 1) & (HIST_RING - 1));
            d.hist[hi] = e.b;
            d.hist_hash[hi] = e.hash;
            d.rec_moves[(size_t)g * d.MAXPLY + p] = (u16)mv;
            d.cur[g] = e.b;
            d.game[g].ply = p + 1;
            d.game[g].game_result = (int8_t)e.result;
            d.game[g].root_dead = 1;
        }
    }
}

// ---- Game seam kernels -------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_set_positions(Dev d, const Board *in, const uint8_t *mask,
                                                      int n, int use_start)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    if (r >= n || (mask && !mask[r])) return;
    Board b;
    if (use_start) {
        b.bb[PAWN] = 0x00FF00000000FF00ull; b.bb[KNIGHT] = 0x4200000000000042ull;
        b.bb[BISHOP] = 0x2400000000000024ull; b.bb[ROOK] = 0x8100000000000081ull;
        b.bb[QUEEN] = 0x0800000000000008ull; b.bb[KING] = 0x1000000000000010ull;
        b.white = 0xFFFFull;
        b.state = mk_state(1, 15, NO_EP, 0, 0);
        b.pad = 0;
    } else {
        b = in[r];
        b.state &= 0xFFFFFu;
        b.pad = 0;
    }
    if (lane == 0) d.game[g].ply = 0;
    PosEval e = eval_position(d, g, b, 0, 0, -1, lane, s);
    if (lane == 0) {
        d.cur[g] = e.b;
        d.hist[(size_t)g * HIST_RING] = e.b;
        d.hist_hash[(size_t)g * HIST_RING] = e.hash;
        d.game[g].game_result = (int8_t)e.result;
        d.game[g].root_dead = 1;
        d.game[g].leaf_kind = LEAF_NONE;
    }
}

__global__ __launch_bounds__(64) void k_legal_moves(Dev d, u16 *moves, int32_t *counts)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    Board b = d.cur[g];
    MoveGenInfo mi = wave_movegen(b, lane, s.mv);
    __syncthreads();
    for (int j = lane; j < mi.n; j += 64) moves[(size_t)r * MAX_MOVES + j] = s.mv[j];
    if (lane == 0) counts[r] = mi.n;
}

__global__ __launch_bounds__(64) void k_push(Dev d, const u16 *moves, uint8_t *ok)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    const u32 mv = moves[r];
    if (mv == NO_MOVE) { if (lane == 0) ok[r] = 0; return; }
    Board b = d.cur[g];
    MoveGenInfo mi = wave_movegen(b, lane, s.mv);
    __syncthreads();
    bool found = false;
    for (int j = lane; j < mi.n; j += 64) found = found || s.mv[j] == mv;
    const bool legal = __ballot(found) != 0;
    __syncthreads();
    if (lane == 0) ok[r] = legal ? 1 : 0;
    if (legal) game_push(d, g, b, mv, lane, s);
}

// DatasetGame.loads / augment_game (dataset.py:21-57): replay a recorded move list, each move through
// the same legality test as Game.move; one launch for every game's whole list.
__global__ __launch_bounds__(64) void k_push_seq(Dev d, const u16 *seq, const int32_t *counts, int stride,
                                                   int32_t *pushed)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    const int n = counts[r] < stride ? counts[r] : stride;
    int i = 0;
    for (; i < n; i++) {
        const u32 mv = seq[(size_t)r * stride + i];
        if (mv == NO_MOVE) break;
        Board b = d.cur[g];
        MoveGenInfo mi = wave_movegen(b, lane, s.mv);
        __syncthreads();
        bool found = false;
        for (int j = lane; j < mi.n; j += 64) found = found || s.mv[j] == mv;
        const bool legal = __ballot(found) != 0;
        __syncthreads();
        if (!legal) break;
        game_push(d, g, b, mv, lane, s);
        __threadfence_block();               // lane 0 wrote cur / ply / history ring: the next ply reads them
        __syncthreads();
    }
    if (lane == 0) pushed[r] = i;
}

// len(Game) and Game.get_result() of every slot of the window, as contiguous arrays for the host
__global__ __launch_bounds__(64) void k_game_scalars(Dev d, int32_t *plies, int8_t *results)
{
    const int r = blockIdx.x, g = r + d.g0;
    if (threadIdx.x == 0) { plies[r] = d.game[g].ply; results[r] = d.game[g].game_result; }
}

__global__ __launch_bounds__(64) void k_encode_cur(Dev d, void *planes)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    Board b = d.cur[g];
    const int p = d.game[g].ply;
    encode_position(d, g, b, 0, p, p, lane, s, planes, r);
}

__global__ __launch_bounds__(64) void k_greedy(Dev d, const float *pol, const uint8_t *mask,
                                               int push, u16 *moves_out)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    if (lane == 0) moves_out[r] = NO_MOVE;
    if (mask && !mask[r]) return;
    Board b = d.cur[g];
    MoveGenInfo mi = wave_movegen(b, lane, s.mv);
    __syncthreads();
    if (mi.n == 0) return;                    // legal[argmax([])] would raise; game is over
    const int bi = argmax_policy(d, r, s.mv, mi.n, pol, lane, FMT_FULL);    // always a full policy
    const u32 mv = s.mv[bi];
    __syncthreads();
    if (lane == 0) moves_out[r] = (u16)mv;
    if (push && d.game[g].game_result == RESULT_NONE) game_push(d, g, b, mv, lane, s);
}

// Game.get_copy (game.py:79-80): deep copy incl. the move stack, slot src -> slot dst
__global__ __launch_bounds__(64) void k_copy_game(Dev d, int dst, int src)
{
    const int lane = threadIdx.x;
    const int p = d.game[src].ply;
    for (int i = lane; i < HIST_RING; i += 64) {
        d.hist[(size_t)dst * HIST_RING + i] = d.hist[(size_t)src * HIST_RING + i];
        d.hist_hash[(size_t)dst * HIST_RING + i] = d.hist_hash[(size_t)src * HIST_RING + i];
    }
    for (int i = lane; i < p; i += 64)
        d.rec_moves[(size_t)dst * d.MAXPLY + i] = d.rec_moves[(size_t)src * d.MAXPLY + i];
    if (lane == 0) {
        d.cur[dst] = d.cur[src];
        d.game[dst].ply = p;
        d.game[dst].game_result = d.game[src].game_result;
        d.game[dst].root_dead = 1;
        d.game[dst].leaf_kind = LEAF_NONE;
    }
}

// Tree.__init__ (mctree.py:105-109: ``Node(root.get_copy())``) when the caller's Game lives in another
// context (the Game arena) than the search engine: the same deep copy across two contexts of one GPU.
__global__ __launch_bounds__(64) void k_copy_game_across(Dev d, int dst, Dev sd, int src)
{
    const int lane = threadIdx.x;
    const int p = sd.game[src].ply;
    if (p > d.MAXPLY) { if (lane == 0) dev_error(d, DERR_PLY_POOL); return; }
    for (int i = lane; i < HIST_RING; i += 64) {
        d.hist[(size_t)dst * HIST_RING + i] = sd.hist[(size_t)src * HIST_RING + i];
        d.hist_hash[(size_t)dst * HIST_RING + i] = sd.hist_hash[(size_t)src * HIST_RING + i];
    }
    for (int i = lane; i < p; i += 64)
        d.rec_moves[(size_t)dst * d.MAXPLY + i] = sd.rec_moves[(size_t)src * sd.MAXPLY + i];
    if (lane == 0) {
        d.cur[dst] = sd.cur[src];
        d.game[dst].ply = p;
        d.game[dst].game_result = sd.game[src].game_result;
        d.game[dst].root_dead = 1;
        d.game[dst].leaf_kind = LEAF_NONE;
    }
}

// ---- SelfPlayTree seam kernels -------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_search_begin(Dev d, void *planes)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    const size_t nb = (size_t)g * d.N, eb = (size_t)g * d.ECAP;
    if (lane == 0) { d.game[g].leaf_kind = LEAF_NONE; d.game[g].path_len = 0; }
    if (d.game[g].game_result != RESULT_NONE) {
        if (lane == 0) { d.game[g].root_dead = 1; d.game[g].n_nodes = 0; d.game[g].root_visits = 0; }
        return;
    }
    Board b = d.cur[g];
    const int p = d.game[g].ply;
    MoveGenInfo mi = wave_movegen(b, lane, s.mv);
    __syncthreads();
    init_edges(d, eb, 0, mi.n, s.mv, lane);
    if (lane == 0) {
        NodeMeta m;
        m.edge0 = 0; m.nmoves = (u16)mi.n; m.nexp = 0; m.result = RESULT_NONE; m.has_s2 = 1;
        m.parent = 0; m.parent_edge = -1;
        d.node[nb].meta = m;
        d.node[nb].s2 = b;
        d.node[nb].h2 = board_hash(b);
        d.game[g].n_nodes = 1;
        d.game[g].edge_top = mi.n;
        d.game[g].root_visits = 1;                // Tree.__init__: root.visits = 1
        d.game[g].root_dead = 0;
        d.path_node[nb] = 0;
    }
    __syncthreads();
    encode_position(d, g, b, 0, p, p, lane, s, planes, r);
}

__global__ __launch_bounds__(64) void k_root_priors(Dev d, const float *pol)
{
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    if (d.game[g].root_dead) return;
    NodeMeta m = d.node[(size_t)g * d.N].meta;
    gather_priors(d, r, (size_t)g * d.ECAP, m.edge0, m.nmoves, pol, lane, FMT_FULL);   // the root's policy is always full
    if (lane == 0) d.counters[(size_t)g * CNT_N + CNT_EVALS] += 1;
}

// simulate + backprop (+ priors of the new node's future children) for the pending simulation
__device__ inline void backup_pending(const Dev &d, int g, int row, int lane, const float *pol2,
                                      const float *val2)
{
    const int kind = uni(d.game[g].leaf_kind);
    if (kind == LEAF_NONE) return;
    const size_t nb = (size_t)g * d.N, eb = (size_t)g * d.ECAP;
    if (kind == LEAF_NEW_REPLY) { dev_error(d, DERR_STATE); return; }
    const int leaf = uni(d.game[g].leaf_node);
    NodeMeta m = d.node[nb + leaf].meta;
    double v;
    unsigned long long evals = 0;
    if (m.result != RESULT_NONE) {
        v = (double)m.result;                          // state.get_result() (mctree.py:268)
    } else {
        v = (double)val2[row];                           // python float of the f32 value head
        gather_priors(d, row, eb, m.edge0, m.nmoves, pol2, lane, d.policy_fmt, d.stats_s2);
        evals = 1;                                     // policy/value(S2)
    }
    if (kind == LEAF_NEW_S2) evals += 1;               // policy(S1) chose the reply
    const int plen = uni(d.game[g].path_len);
    for (int l = lane; l < plen; l += 64) {
        Edge *e = d.edge + eb + d.path_edge[nb + l];
        e->visits += 1;
        e->value = __dadd_rn(e->value, v);
    }
    if (lane == 0) {
        d.game[g].root_visits += 1;
        d.game[g].leaf_kind = LEAF_NONE;
        unsigned long long *c = d.counters + (size_t)g * CNT_N;
        c[CNT_SIMS] += 1;
        c[CNT_DEPTH] += plen;
        c[CNT_EVALS] += evals;
        if (kind == LEAF_TERMINAL_HIT) c[CNT_TERMINAL] += 1;
        else { c[CNT_NODES] += 1; c[CNT_BRANCH] += m.nmoves; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_backup(Dev d, const float *pol2, const float *val2)
{
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    if (d.game[g].root_dead) return;
    backup_pending(d, g, r, lane, pol2, val2);
}

__global__ __launch_bounds__(64, 4) void k_select_expand(Dev d, const float *pol2, const float *val2,
                                                      void *planes1)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    if (lane == 0) d.lab_n1[r] = 0;                     // no policy(S1) wanted unless a reply is needed
    if (d.game[g].root_dead) return;
    backup_pending(d, g, r, lane, pol2, val2);

    const size_t nb = (size_t)g * d.N, eb = (size_t)g * d.ECAP;
    const bool legacy = (d.flags & 1u) != 0;
    const int p = uni(d.game[g].ply);
    int node = 0, level = 0;
    int edge0 = 0, nmoves = 0, nexp = 0, result = RESULT_NONE, parent_edge = -1;
    bool need_meta = true;                                             // false: the parent's edge record said it all
    if (d.ECAP >= (1 << HINT_EDGE_BITS)) { dev_error(d, DERR_EDGE_POOL); return; }   // edge slots fit the hint
    for (;;) {
        if (need_meta) {
            NodeMeta m = d.node[nb + node].meta;
            edge0 = uni(m.edge0); nmoves = uni(m.nmoves); nexp = uni(m.nexp);
            result = uni(m.result); parent_edge = uni(m.parent_edge);
        }
        if (result != RESULT_NONE) {                                   // is_terminal_state
            if (lane == 0) { d.game[g].leaf_kind = LEAF_TERMINAL_HIT; d.game[g].leaf_node = node; }
            break;
        }
        if (nexp < nmoves) {                                           // not fully expanded
            const int j = nmoves - 1 - nexp;                           // list.pop(): last first
            const int edge = edge0 + j;
            const u32 mv = d.edge[eb + edge].move;
            const int c = uni(d.game[g].n_nodes);                           // wave-uniform: scalar addressing
            if (c >= d.N || level + 1 >= d.N) { dev_error(d, DERR_NODE_POOL); break; }
            if (lane == 0) {
                d.node[nb + node].meta.nexp = (u16)(nexp + 1);
                if (nexp + 1 == nmoves && parent_edge >= 0)            // fully expanded from now on: leave the hint
                    d.edge[eb + parent_edge].pad = hint_pack(edge0, nmoves);
                d.path_edge[nb + level] = edge;
                d.path_node[nb + level + 1] = (u16)c;
                d.game[g].n_nodes = c + 1;
            }
            level++;
            Board parent = d.node[nb + node].s2;
            Board s1 = apply_move(parent, mv);
            __syncthreads();                                           // path_node visible
            PosEval e = eval_position(d, g, s1, 2 * level - 1, p, p, lane, s);
            NodeMeta cm;
            cm.edge0 = 0; cm.nmoves = (u16)e.n; cm.nexp = 0; cm.result = (int8_t)e.result;
            cm.has_s2 = 0; cm.parent = (u16)node; cm.parent_edge = edge;
            if (lane == 0) {
                d.node[nb + c].s1 = e.b;
                d.node[nb + c].h1 = e.hash;
                d.node[nb + c].meta = cm;
                d.game[g].leaf_node = c;
            }
            if (e.result != RESULT_NONE) {                              // game ended on our move
                if (lane == 0) {
                    d.node[nb + c].s2 = e.b;
                    d.node[nb + c].h2 = e.hash;
                    d.edge[eb + edge].child = (u16)(c | CHILD_TERMINAL);
                    d.game[g].leaf_kind = LEAF_NEW_S1_OVER;
                }
            } else {
                for (int i = lane; i < e.n; i += 64) d.s1_moves[(size_t)g * MAX_MOVES + i] = s.mv[i];
                if (d.policy_fmt) write_labels(d, r, s.mv, e.n, d.lab_s1, d.lab_n1, lane);
                if (lane == 0) {
                    d.edge[eb + edge].child = (u16)c;
                    d.game[g].s1_n = e.n;
                    d.game[g].leaf_kind = LEAF_NEW_REPLY;
                }
                __syncthreads();
                encode_position(d, g, e.b, 2 * level - 1, p, p, lane, s, planes1, r);
            }
            break;
        }
        // ---- get_best_child (mctree.py:89-95): argmax of Q+U, first max in children order,
        // i.e. the LARGEST legal index among equals
        double best = -__builtin_inf();
        int bj = -1, bchild = 0;
        u32 bhint = 0;
        for (int base = 0; base < nmoves; base += 64) {
            const int j = base + lane;
            if (j < nmoves) {
                const Edge e = d.edge[eb + edge0 + j];                 // one 24-byte record per lane
                const int n = e.visits;
                const bool term = (e.child & CHILD_TERMINAL) != 0;
                const double den = (double)(1 + n);
                const double q = __ddiv_rn(e.value, den);
                const double sumv = term ? 0.0 : (double)(n - 1);
                const double cp = legacy ? __dmul_rn(10.0, (double)e.prior)
                                         : (double)__fmul_rn(10.0f, e.prior);
                const double u = __dmul_rn(cp, __ddiv_rn(__dsqrt_rn(sumv), den));
                const double sc = __dadd_rn(q, u);
                if (bj < 0 || sc > best || (sc == best && j > bj)) { best = sc; bj = j; bchild = e.child; bhint = e.pad; }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oj = __shfl_xor(bj, o);
            const int oc = __shfl_xor(bchild, o);
            const u32 oh = __shfl_xor(bhint, o);
            const bool take = oj >= 0 && (bj < 0 || ob > best || (ob == best && oj > bj));
            if (take) { best = ob; bj = oj; bchild = oc; bhint = oh; }
        }
        bj = uni(bj);
        const int edge = edge0 + bj;
        const int child = uni(bchild) & CHILD_NONE;                    // the winner's record held it
        if (level + 1 >= d.N) { dev_error(d, DERR_NODE_POOL); break; }
        if (lane == 0) {
            d.path_edge[nb + level] = edge;
            d.path_node[nb + level + 1] = (u16)child;
        }
        level++;
        node = uni(child);
        const u32 hint = uni(bhint);
        if (uni(bchild) & CHILD_TERMINAL) {                            // the child's state ended the game: as its
            need_meta = false;                                         // record would say (result != RESULT_NONE)
            result = 0;                                                // any value but RESULT_NONE: only tested
        } else if (hint & HINT_FULL) {                                 // fully expanded: straight to its edge records
            need_meta = false;
            edge0 = (int)(hint & ((1u << HINT_EDGE_BITS) - 1)); nmoves = (int)((hint >> HINT_EDGE_BITS) & 0xFFu); nexp = nmoves;
            result = RESULT_NONE;
        } else {
            need_meta = true;
        }
    }
    if (lane == 0) d.game[g].path_len = level;
}

__global__ __launch_bounds__(64) void k_reply(Dev d, const float *pol1, void *planes2)
{
    __shared__ WaveLds s;
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    if (lane == 0) d.lab_n2[r] = 0;                     // no policy(S2) wanted unless a new node gets priors
    if (d.game[g].root_dead || d.game[g].leaf_kind != LEAF_NEW_REPLY) return;
    const size_t nb = (size_t)g * d.N, eb = (size_t)g * d.ECAP;
    const int c = uni(d.game[g].leaf_node), level = uni(d.game[g].path_len), n1 = uni(d.game[g].s1_n), p = uni(d.game[g].ply);
    Board s1 = d.node[nb + c].s1;
    // agent.best_move(S1, real_game=True): legal[argmax(policy masked to legal)]
    const u16 *mv1 = d.s1_moves + (size_t)g * MAX_MOVES;
    const int bi = argmax_policy(d, r, mv1, n1, pol1, lane, d.policy_fmt, d.stats_s1);
    const u32 reply = mv1[bi];
    Board s2 = apply_move(s1, reply);
    PosEval e = eval_position(d, g, s2, 2 * level, p, p, lane, s);
    const int edge0 = uni(d.game[g].edge_top);
    if (edge0 + e.n > d.ECAP) { dev_error(d, DERR_EDGE_POOL); return; }
    init_edges(d, eb, edge0, e.n, s.mv, lane);
    if (d.policy_fmt && e.result == RESULT_NONE) write_labels(d, r, s.mv, e.n, d.lab_s2, d.lab_n2, lane);
    if (lane == 0) {
        NodeMeta m = d.node[nb + c].meta;
        m.edge0 = edge0; m.nmoves = (u16)e.n; m.nexp = 0; m.result = (int8_t)e.result; m.has_s2 = 1;
        d.node[nb + c].meta = m;
        d.node[nb + c].s2 = e.b;
        d.node[nb + c].h2 = e.hash;
        d.node[nb + c].reply = (u16)reply;
        d.game[g].edge_top = edge0 + e.n;
        if (e.result != RESULT_NONE) d.edge[eb + m.parent_edge].child = (u16)(c | CHILD_TERMINAL);
        d.game[g].leaf_kind = LEAF_NEW_S2;
    }
    __syncthreads();
    encode_position(d, g, e.b, 2 * level, p, p, lane, s, planes2, r);
}

__global__ __launch_bounds__(64) void k_root_children(Dev d, int32_t *nchild, int32_t *visits,
                                                      double *values, float *priors, u16 *moves,
                                                      u16 *replies, int32_t *root_visits)
{
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    const size_t nb = (size_t)g * d.N, eb = (size_t)g * d.ECAP;
    if (d.game[g].root_dead) {
        if (lane == 0) { nchild[r] = 0; root_visits[r] = 0; }
        return;
    }
    NodeMeta m = d.node[nb].meta;
    if (lane == 0) { nchild[r] = m.nexp; root_visits[r] = d.game[g].root_visits; }
    for (int k = lane; k < m.nexp; k += 64) {
        const Edge e = d.edge[eb + m.edge0 + (m.nmoves - 1 - k)];   // children order = reverse legal
        const size_t o = (size_t)r * MAX_MOVES + k;
        const int c = e.child & CHILD_NONE;
        visits[o] = e.visits;
        values[o] = e.value;
        // _update_prior runs only once the node is fully expanded (mctree.py:254-255); until
        // then the reference's children still carry Node.prior = 1
        priors[o] = m.nexp < m.nmoves ? 1.0f : e.prior;
        moves[o] = e.move;
        replies[o] = d.node[nb + c].meta.has_s2 ? d.node[nb + c].reply : NO_MOVE;
    }
}

__global__ __launch_bounds__(64) void k_advance(Dev d, const int32_t *chosen, u16 *bm, u16 *am)
{
    const int r = blockIdx.x, g = r + d.g0, lane = threadIdx.x;
    if (lane != 0) return;
    bm[r] = NO_MOVE; am[r] = NO_MOVE;
    const int k = chosen[r];
    if (k < 0 || d.game[g].root_dead) return;
    const size_t nb = (size_t)g * d.N, eb = (size_t)g * d.ECAP;
    NodeMeta m = d.node[nb].meta;
    if (k >= m.nexp || d.game[g].leaf_kind != LEAF_NONE) { dev_error(d, DERR_STATE); return; }
    const Edge ed = d.edge[eb + m.edge0 + (m.nmoves - 1 - k)];
    const int c = ed.child & CHILD_NONE;
    NodeMeta cm = d.node[nb + c].meta;
    const int p = d.game[g].ply;
    const int np = p + (cm.has_s2 ? 2 : 1);
    if (np > d.MAXPLY) { dev_error(d, DERR_PLY_POOL); return; }
    size_t hi = (size_t)g * HIST_RING + ((p + 1) & (HIST_RING - 1));
    d.hist[hi] = d.node[nb + c].s1;
    d.hist_hash[hi] = d.node[nb + c].h1;
    d.rec_moves[(size_t)g * d.MAXPLY + p] = ed.move;
    bm[r] = ed.move;
    if (cm.has_s2) {
        hi = (size_t)g * HIST_RING + ((p + 2) & (HIST_RING - 1));
        d.hist[hi] = d.node[nb + c].s2;
        d.hist_hash[hi] = d.node[nb + c].h2;
        d.rec_moves[(size_t)g * d.MAXPLY + p + 1] = d.node[nb + c].reply;
        am[r] = d.node[nb + c].reply;
    }
    d.cur[g] = d.node[nb + c].s2;                 // node state (S1 copy when the game ended there)
    d.game[g].ply = np;
    d.game[g].game_result = cm.result;
    d.game[g].root_dead = 1;                       // the tree is consumed: fresh tree per move
}

}  // namespace crl
