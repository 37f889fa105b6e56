// state.hpp -- HBM layout of the lockstep self-play state (one context per GPU).
//
// Everything is structure-of-arrays, game-major, sized at crl_create and never
// reallocated.  G games advance in lockstep; each owns a fresh search tree per
// move (agentdistributed.py:61-63) of at most N = max_sims + 1 nodes, because
// one simulation creates at most one node (mctree.py:231-257).
//
//   games   game[G]               32-byte row of per-game scalars (ply, result, tree counters,
//                                 pending-simulation state)
//           cur[G]                current position (== hist ring at ply)
//           hist[G][256]          ring of the last 256 positions  (encoder history,
//           hist_hash[G][256]     fivefold repetition; python-chess move stack)
//           rec_moves[G][P]       the game record, u16 move ids (game.py:59-66)
//   nodes   node[G][N]            176-byte record: 16-byte meta (edge range, expansion cursor,
//                                 result), filter hashes and boards of S1 (after our move) and
//                                 S2 (after the stored reply), the reply -- one record so that a
//                                 new node is a few contiguous sectors and the kernels hold ONE
//                                 base pointer instead of six (scalar-register pressure spilled
//                                 26 VGPRs of k_select_expand in the array-per-field layout)
//   edges   edge[G][ECAP]         one 24-byte record per LEGAL MOVE of a node, in python-chess
//                                 order: child's value sum (f64), visits (i32), prior (f32),
//                                 move (u16), child node id | terminal<<15.  A node's records
//                                 are contiguous: the PUCT scan of get_best_child
//                                 (mctree.py:89-95) is one coalesced read of 64 x 24 bytes, the
//                                 backup's visits/value update touches one sector per level,
//                                 and the winner's child id comes out of the same record.
//   step    path_*[G][N]          the selection path of the pending simulation
#pragma once
#include "board.hpp"

namespace crl {

constexpr int HIST_RING = 256;
constexpr int MAX_MOVES = 256;
constexpr int MAX_BRANCH = 218;      // most legal moves in any reachable position
constexpr int N_LABELS = 1968;
constexpr int PLANES = 128;
constexpr u16 CHILD_NONE = 0x7FFF;
constexpr u16 CHILD_TERMINAL = 0x8000;

enum LeafKind : uint8_t {
    LEAF_NONE = 0,          // no simulation pending
    LEAF_TERMINAL_HIT = 1,  // selection ended on an existing terminal node
    LEAF_NEW_S1_OVER = 2,   // new node whose game ended on our move (state = S1)
    LEAF_NEW_REPLY = 3,     // new node waiting for the opponent's reply (needs policy(S1))
    LEAF_NEW_S2 = 4         // new node complete (state = S2)
};

struct __attribute__((aligned(16))) NodeMeta {
    int32_t edge0;          // first edge slot (game-relative)
    u16 nmoves;             // b = number of legal moves = number of edge slots
    u16 nexp;               // children created so far; next expansion is legal index b-1-nexp
    int8_t result;          // Game.get_result() of the node state; RESULT_NONE = running
    uint8_t has_s2;         // state is S2 (1) or S1 (0, game ended on our move)
    u16 parent;             // parent node id
    int32_t parent_edge;    // edge slot in the parent (game-relative), -1 for the root
};
static_assert(sizeof(NodeMeta) == 16, "NodeMeta must be 16 bytes");

struct __attribute__((aligned(16))) NodeRow {
    NodeMeta meta;
    u64 h1, h2;             // transposition-key filter hashes of S1 / S2
    Board s1, s2;           // S1 (after our move) / S2 (after the stored reply; S1 copy if the game ended there)
    u16 reply;              // the stored reply
    u16 pad[7];
};
static_assert(sizeof(NodeRow) == 176, "NodeRow must be 176 bytes");

struct __attribute__((aligned(16))) GameRow {
    int32_t ply;            // len(move_stack)
    int32_t n_nodes, edge_top, root_visits;              // the tree of the current move
    int32_t path_len, leaf_node, s1_n;                   // the pending simulation
    int8_t game_result;     // Game.get_result(); RESULT_NONE = running
    uint8_t root_dead;      // no live tree (finished game, or the tree was consumed by crl_advance)
    uint8_t leaf_kind;      // LeafKind of the pending simulation
    uint8_t pad;
};
static_assert(sizeof(GameRow) == 32, "GameRow must be 32 bytes");

// One legal move of a node = one child slot (Node.value / visits / prior of mctree.py:28-37).
struct __attribute__((aligned(8))) Edge {
    double value;           // child's value sum
    int32_t visits;         // child's visit count
    float prior;            // child's prior (1 until the parent is fully expanded)
    u16 move;               // the move (u16 id)
    u16 child;              // child node id | CHILD_TERMINAL, CHILD_NONE while unexpanded
    uint32_t pad;           // descent hint (search.hpp: hint_pack): 0, or -- once the CHILD is fully expanded --
                            // HINT_FULL | the child's nmoves << 23 | the child's edge0: select then goes on to the
                            // child's edge records without reading the child's node record first
};
static_assert(sizeof(Edge) == 24, "Edge must be 24 bytes");

enum Counter { CNT_SIMS = 0, CNT_NODES, CNT_DEPTH, CNT_BRANCH, CNT_EVALS, CNT_TERMINAL, CNT_N };

struct Dev {
    int G, N, ECAP, MAXPLY;
    int g0;                            // first slot of the active window (I/O rows are window-relative)
    u32 flags;
    int plane_fmt;                     // 0: encoders write fp16 NHWC planes, 1: 128 bit planes per position
    int policy_fmt;                    // 0: evaluators hand back policy[row][1968]; 1: priors[row][256] of the
                                       //    legal moves the search kernels listed in lab_s1 / lab_s2; 2: the same
                                       //    rows holding LOGITS, to be normalised on read with the slice
                                       //    statistics in stats_s1 / stats_s2 (csrc/slices.hpp)
    const float2 *stats_s1, *stats_s2; // [rows][8] (max, sum exp) per label slice, window-relative rows
    // games
    GameRow *game;
    Board *cur;
    Board *hist;
    u64 *hist_hash;
    u16 *rec_moves;
    // tree
    NodeRow *node;
    Edge *edge;
    // pending simulation
    int32_t *path_edge;
    u16 *path_node;
    u16 *s1_moves;
    // policy labels of the legal moves of the position each tower call evaluates, in legal order
    // (rows are window-relative like every evaluator buffer): S1 by k_select_expand, S2 by k_reply
    u16 *lab_s1, *lab_s2;              // [G][MAX_MOVES]
    int32_t *lab_n1, *lab_n2;          // [G]; 0 when the row needs no policy this step
    // misc
    const u16 *lut;                    // [5][4096] move -> label index (0xFFFF = none)
    unsigned long long *counters;      // [G][CNT_N]
    int32_t *err;                      // sticky device error code
};

enum DevErr { DERR_NONE = 0, DERR_NODE_POOL = 1, DERR_EDGE_POOL = 2, DERR_PLY_POOL = 3,
              DERR_STATE = 4, DERR_BRANCH = 5, DERR_LABEL = 6 };

__device__ inline int label_of(const Dev &d, u32 mv)
{
    u32 p = (mv >> 12) & 7;
    u32 slot = p ? 6 - p : 0;          // Q,R,B,N -> 1,2,3,4
    return d.lut[slot * 4096 + (mv & 4095)];
}

}  // namespace crl
