/*
 * chessrl_hip.h -- C-ABI of libchessrl_hip.so, the MI355X (gfx950) drop-in for
 * the self-play MCTS simulation loop of AIRLegend/ChessRL.
 *
 * The reference is pure Python and defines no FFI; its seams for this path are
 * duck-typed Python methods.  Each entry point below
 * names the reference method(s) it replaces (paths relative to
 * /root/reference/src/chessrl/).  The Python host mirror (chessrl_amd/game.py,
 * agent.py, mctree.py, selfplay.py) binds these with ctypes; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes, no torch types; every call returns 0
 * on success or a negative crl_status and records a message retrievable with
 * crl_last_error(); one context per GPU, not thread-safe per context; "dev_"
 * pointers are device memory (e.g. torch.Tensor.data_ptr()), all others are
 * caller-owned host memory; kernels are enqueued on the stream given to
 * crl_set_stream() (torch's current stream) so they order with the tower and
 * can be captured into a hipGraph; calls that fill host buffers synchronise
 * that stream themselves.  G = max_games given to crl_create.
 */
#ifndef CHESSRL_HIP_H
#define CHESSRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One chess position, 64 bytes.  Squares a1 = 0 ... h8 = 63 (python-chess). */
typedef struct crl_board {
    uint64_t bb[6];   /* pawns, knights, bishops, rooks, queens, kings (both colours) */
    uint64_t white;   /* squares occupied by white pieces                              */
    uint32_t state;   /* bit 0      side to move (1 = white)
                         bits 1-4   castling rights: WK(h1) WQ(a1) BK(h8) BQ(a8)
                         bits 5-11  en-passant square, 64 = none (set on EVERY double push)
                         bits 12-19 halfmove clock (saturates at 255)
                         bit 20     a legal en-passant capture exists (derived; recomputed
                                    by the library, ignored on input)                   */
    uint32_t pad;
} crl_board;

/* A move is a uint16: from | to << 6 | promo << 12, promo in {0, 2=N, 3=B, 4=R, 5=Q};
 * castling is the king move (e1g1), as python-chess prints it.  0xFFFF = "no move". */
#define CRL_NO_MOVE 0xFFFFu
#define CRL_MAX_MOVES 256      /* row stride of every per-position move list           */
#define CRL_N_LABELS 1968      /* netencoder.get_uci_labels(), netencoder.py:94-134     */
#define CRL_PLANES 128         /* 127 reference planes + 1 zero pad, fp16 NHWC          */
#define CRL_RESULT_NONE 2      /* Game.get_result() is None                             */

typedef enum crl_status {
    CRL_OK = 0,
    CRL_ERR_ARG = -1,          /* bad argument                                          */
    CRL_ERR_HIP = -2,          /* HIP runtime failure (message has the hipError string) */
    CRL_ERR_CAPACITY = -3,     /* a device pool overflowed (nodes / edges / plies)      */
    CRL_ERR_STATE = -4         /* call not valid in the current search state            */
} crl_status;

/* crl_create flags */
#define CRL_FLAG_NUMPY_LEGACY 1u  /* PUCT: 10*prior evaluated in float64 (numpy 1.x scalar
                                     promotion, the reference's pinned numpy==1.17.2);
                                     default is numpy>=2 (product rounded to float32).  */

typedef struct crl_ctx crl_ctx;

/* ---- identity of the library ------------------------------------------------------------- */
/* Bumped whenever a signature or the meaning of an argument below changes.  A binding calls
 * crl_abi_version() first and refuses a library of another version (a ctypes call through a stale
 * signature hands the GPU garbage pointers).  crl_source_hash(): sha256 over the sources (csrc/ + this
 * header + compiler flags) the library was built from, as chessrl_amd/_lib.py computes it; the string
 * is also findable in the file itself behind the marker "CRL_SRC_HASH=".  No reference counterpart. */
#define CRL_ABI_VERSION 8
int  crl_abi_version(void);
const char *crl_source_hash(void);

/* ---- lifetime ------------------------------------------------------------------- */
int  crl_create(crl_ctx **out, int device, int max_games, int max_sims, int max_plies,
                uint32_t flags);
void crl_destroy(crl_ctx *ctx);
int  crl_set_stream(crl_ctx *ctx, void *hip_stream);
int  crl_sync(crl_ctx *ctx);                 /* wait for the stream; report device errors */
const char *crl_last_error(crl_ctx *ctx);    /* ctx may be NULL for crl_create failures   */
/* Restrict every later call to slots [first, first+count): host arrays and dev_ rows are then
 * indexed relative to `first` and sized by `count` instead of G (default window = all G). */
int  crl_set_window(crl_ctx *ctx, int first, int count);
/* What the encoding entry points (crl_encode, crl_search_begin, crl_sim_select_expand,
 * crl_sim_reply: netencoder.get_game_state, netencoder.py:72-91) write per position:
 * CRL_PLANES_F16 (default) = the fp16 NHWC planes [8][8][128] (16 KiB);
 * CRL_PLANES_BITS = the same information as 128 plane bitboards, uint64 [128] (1 KiB; bit s of
 * plane c = channel c on square s, a1 = 0): the input of crl_trunk_forward_bitplanes, which expands
 * it on chip, so that the planes never exist in HBM. */
#define CRL_PLANES_F16  0
#define CRL_PLANES_BITS 1
int  crl_set_plane_format(crl_ctx *ctx, int format);
/* What the evaluator hands back to crl_sim_select_expand / crl_sim_reply / crl_sim_backup as
 * "policy" (mctree.py:298-303 _update_prior and agentdistributed.py:80-82 read the 1968-vector only
 * at the labels of the position's legal moves):
 * CRL_POLICY_FULL (default) = dev_policy rows are float32 [1968], the kernels gather at the labels;
 * CRL_POLICY_LEGAL = dev_policy rows are float32 [CRL_MAX_MOVES]: entry j is the probability of the
 * position's legal move j (crl_legal_moves order).  The search kernels then publish, for the position
 * each tower call evaluates, the labels of its legal moves: crl_eval_labels(ctx, 0, ...) after
 * crl_sim_select_expand (S1), crl_eval_labels(ctx, 1, ...) after crl_sim_reply (S2): device pointers
 * to uint16 [count][CRL_MAX_MOVES] and int32 [count] (0 = this row needs no policy in this step),
 * rows window-relative, valid for the life of the context.  crl_heads_forward_legal consumes them.
 * crl_search_root_priors and crl_greedy_moves always take full policies.
 * CRL_POLICY_LEGAL_RAW = rows as CRL_POLICY_LEGAL, but holding the LOGITS of the legal moves, with the
 * softmax statistics of the board (8 label slices x (max, sum exp), float [rows][16]) at the address given
 * to crl_set_policy_stats(ctx, which, ...) (which = 0: the evaluation of S1, read by crl_sim_reply; 1: of
 * S2, read by crl_sim_select_expand / crl_sim_backup): the kernels form exp(l - M) / S on read -- the
 * arithmetic of the heads' normalising pass, same bits -- so small batches need no such pass
 * (crl_heads_forward_legal_raw).  Both addresses must be set before the format is selected. */
#define CRL_POLICY_FULL  0
#define CRL_POLICY_LEGAL 1
#define CRL_POLICY_LEGAL_RAW 2
int  crl_set_policy_format(crl_ctx *ctx, int format);
int  crl_set_policy_stats(crl_ctx *ctx, int which, const void *dev_stats_f32);
int  crl_eval_labels(crl_ctx *ctx, int which, const uint16_t **dev_labels, const int32_t **dev_counts);
int  crl_max_games(crl_ctx *ctx);
int  crl_max_sims(crl_ctx *ctx);

/* netencoder.get_uci_labels (netencoder.py:94-134): move id of each of the 1968 labels */
int  crl_uci_label_moves(uint16_t *moves_out /*1968*/);

/* ---- Game seam (game.py) ------------------------------------------------------------ */
/* Game.__init__/reset (game.py:17-26,82-83): slots with mask[g] != 0 (NULL = all)
 * restart from the standard position with an empty move stack. */
int  crl_reset_games(crl_ctx *ctx, const uint8_t *mask);
/* Game(board=...) / set_fen (game.py:17-21,71-72): slots 0..n-1 take `boards` with an
 * EMPTY move stack (tests: arbitrary positions). */
int  crl_set_positions(crl_ctx *ctx, const crl_board *boards, int n);
int  crl_get_positions(crl_ctx *ctx, crl_board *boards_out, int n);
/* Game.get_copy (game.py:79-80): slot dst becomes a deep copy (board AND move stack) of
 * slot src; absolute slot numbers, independent of the window. */
int  crl_copy_game(crl_ctx *ctx, int dst, int src);
/* Tree.__init__ (mctree.py:105-109, ``Node(root.get_copy())``) and Game(board=other.board) when the
 * two games live in different contexts of one GPU (the Game arena and a search engine): slot dst
 * of ctx becomes a deep copy of slot src of src_ctx -- position, move stack and the history the
 * encoder and the repetition rule read -- whatever position the source game started from.
 * The record must fit ctx's max_plies (CRL_ERR_CAPACITY otherwise).  Absolute slot numbers. */
int  crl_copy_game_from(crl_ctx *ctx, int dst, crl_ctx *src_ctx, int src);
/* Game.get_legal_moves (game.py:43-57): python-chess generation order.  moves may be NULL when
 * only len(get_legal_moves()) is wanted (the 512-byte rows are then not copied back). */
int  crl_legal_moves(crl_ctx *ctx, uint16_t *moves /*G x 256*/, int32_t *counts /*G*/);
/* Game.move (game.py:28-41): applied iff in the legal list; ok[g] = 1/0; CRL_NO_MOVE skips. */
int  crl_push_moves(crl_ctx *ctx, const uint16_t *moves /*G*/, uint8_t *ok /*G*/);
/* DatasetGame.loads / augment_game (dataset.py:21-57): every game replays its own recorded move
 * list moves[g*stride .. g*stride+counts[g]) in ONE launch, each move through Game.move's legality
 * test; pushed[g] = number of moves applied (stops at the first illegal move or CRL_NO_MOVE). */
int  crl_push_sequences(crl_ctx *ctx, const uint16_t *moves /*G x stride*/, const int32_t *counts /*G*/,
                        int stride, int32_t *pushed /*G*/);
/* Game.get_result (game.py:92-109): 1 / -1 / 0, CRL_RESULT_NONE while running. */
int  crl_results(crl_ctx *ctx, int8_t *result /*G*/);
/* len(Game) and Game.get_history()['moves'] (game.py:59-66,111-112). */
int  crl_records(crl_ctx *ctx, uint16_t *moves /*G x max_plies or NULL*/, int32_t *plies /*G*/,
                 int8_t *result /*G or NULL*/);

/* ---- encoder seam (netencoder.py) ------------------------------------------------------ */
/* netencoder.get_game_state (netencoder.py:72-91) for every slot: fp16 NHWC
 * [G][8][8][128] (row 0 = rank 8, col 0 = file a; channel 127 is a zero pad). */
int  crl_encode(crl_ctx *ctx, void *dev_planes_f16);

/* ---- Agent seam (agentdistributed.py) ----------------------------------------------- */
/* AgentDistributed.best_move(real_game=True) (agentdistributed.py:56-58): per slot with
 * mask[g] != 0 (NULL = all): legal[argmax(policy[label(m)] for m in legal)], first max.
 * dev_policy_f32 is [G][1968].  If push != 0 the move is also played (selfplay.py:68-70).
 * moves_out (host, may be NULL) receives the chosen move ids (CRL_NO_MOVE if game over). */
int  crl_greedy_moves(crl_ctx *ctx, const void *dev_policy_f32, const uint8_t *mask, int push,
                      uint16_t *moves_out);

/* ---- SelfPlayTree seam (mctree.py), lockstep over all slots, threads=1 semantics -------- */
/* Tree.__init__ (mctree.py:105-111): fresh tree per slot, root = current game, visits 1.
 * Finished games get an inert root.  Also encodes every root into dev_planes_f16 (the
 * reference evaluates the root's policy in _update_prior, mctree.py:298-303). */
int  crl_search_begin(crl_ctx *ctx, void *dev_planes_f16);
/* Root part of _update_prior: priors of the root's children from the tower's policy. */
int  crl_search_root_priors(crl_ctx *ctx, const void *dev_policy_f32);
/* First half of explore_tree (mctree.py:200-257): [backprop of the previous simulation
 * (mctree.py:278-296) if one is pending, using the S2 outputs], then select to a leaf,
 * pop the last unexpanded move, play it (S1) and encode S1 where the opponent must reply.
 * dev_policy_s2/dev_value_s2 may be NULL only when no simulation is pending. */
int  crl_sim_select_expand(crl_ctx *ctx, const void *dev_policy_s2_f32,
                           const void *dev_value_s2_f32, void *dev_planes_s1_f16);
/* Second half of expand (mctree.py:244-250): opponent's greedy reply from policy(S1),
 * S2 = S1 + reply, Node(S2) with its legal moves, encode S2. */
int  crl_sim_reply(crl_ctx *ctx, const void *dev_policy_s1_f32, void *dev_planes_s2_f16);
/* simulate + backprop + child priors (mctree.py:259-303) for the pending simulation. */
int  crl_sim_backup(crl_ctx *ctx, const void *dev_policy_s2_f32, const void *dev_value_s2_f32);
/* [c.visits for c in root.children] etc., CHILDREN order (= reverse legal order);
 * rows of 256; any output may be NULL. */
int  crl_root_children(crl_ctx *ctx, int32_t *nchild /*G*/, int32_t *visits, double *values,
                       float *priors, uint16_t *moves, uint16_t *replies, int32_t *root_visits);
/* selfplay.play_game's two pushes (selfplay.py:77-78) for the chosen root child per slot
 * (children-order index, -1 = leave the slot alone).  bm/am (host, may be NULL) get our
 * move and the stored reply (CRL_NO_MOVE when the game ended on our move). */
int  crl_advance(crl_ctx *ctx, const int32_t *chosen /*G*/, uint16_t *bm, uint16_t *am);
/* The move boundary of the lockstep runner in TWO synchronising calls instead of five (a synchronisation costs
 * ~70 us; at C2 a move is 13 ms).  crl_end_move_fetch = crl_sim_backup + crl_root_children(nchild, visits,
 * root_visits) + len(Game) of every slot; the host then computes the policies (compute_policy, mctree.py:305-322)
 * and crl_advance_fetch = crl_advance + crl_results + the number of legal moves of every NEXT root
 * (len(get_legal_moves()): how many children the next search can have -- the runner draws its Dirichlet
 * noise ahead with it).  Same kernels, same values as the single calls. */
int  crl_end_move_fetch(crl_ctx *ctx, const void *dev_policy_s2_f32, const void *dev_value_s2_f32,
                        int32_t *nchild /*G*/, int32_t *visits /*G x 256*/, int32_t *root_visits /*G*/,
                        int32_t *plies /*G*/);
int  crl_advance_fetch(crl_ctx *ctx, const int32_t *chosen /*G*/, uint16_t *bm, uint16_t *am,
                       int8_t *results /*G*/, int32_t *legal_counts /*G*/);
/* Device-side counters since crl_create: [0] simulations run, [1] nodes created,
 * [2] sum of selection depth (edges), [3] sum of legal moves over created nodes,
 * [4] tower evaluations consumed, [5] terminal leaves hit. */
int  crl_counters(crl_ctx *ctx, uint64_t *out6);

/* ---- tower seam (model.py) -------------------------------------------------------------- */
/* Residual trunk of ChessModel (model.py:33-37,111-122: stem conv + n_blocks residual blocks,
 * BatchNorm folded) for `filters` in {64, 128, 256} (BASELINE configs C2, C3/C4, C5; 256 is the
 * reference's own width, model.py:33) as ONE fused MFMA kernel, plus the three 1x1 head
 * convolutions with their BN + ReLU (model.py:40-42,51-55).  fp16 NHWC planes
 * [n_boards][8][8][128] in.  dev_wtiles_f16 holds the folded fp16 kernels as 64-byte-row planes in
 * consumption order [conv][tap = ky*3+kx][in-ch/32][filters rows][4 chunks][8 in] -- byte for byte the
 * image the kernel keeps in LDS, so that its weight DMA copies contiguous blocks: row r of a
 * plane holds output channel (r & ~31) + 8*((r & 15) >> 2) + 4*((r >> 4) & 1) + (r & 3), and the
 * 16-byte chunk with input channels 8c .. 8c+7 of that row sits at position c ^ ((-(r >> 2)) & 3)
 * (chessrl_amd/model.py:_pack_fused is the reference packer); the stem has 128 input channels, every
 * other conv `filters`; dev_bias_f32 is [1+2*n_blocks][filters].
 * Outputs (either may be NULL): dev_out_f32 = fp32 trunk activations [n_boards][8][8][filters];
 * dev_head_out_f32 = [n_boards][192] floats after ReLU: [0,128) the policy head in Keras Flatten
 * order (position*2 + channel), [128,192) the value head; from dev_head_w_f32 [3][filters] and
 * dev_head_b_f32 [3].  n_boards % 4 == 0.
 * Stateless: needs no crl_ctx. */
int  crl_trunk_forward(void *hip_stream, int filters, const void *dev_planes_f16,
                       const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                       int n_boards, int n_blocks, const void *dev_head_w_f32,
                       const void *dev_head_b_f32, void *dev_head_out_f32);

/* The general form.  flags: CRL_TRUNK_BITPLANES = dev_planes holds plane bitboards (as
 * crl_trunk_forward_bitplanes); CRL_TRUNK_SPLIT = precision mode "f16x3": every operand is carried
 * as two fp16 numbers (hi + lo) and every product is three MFMAs (hi.Whi + hi.Wlo + lo.Whi, fp32
 * accumulation), which puts the outputs within ~1e-5 of an fp32 evaluation of the same weights
 * whatever the weights are (one fp16 MFMA per product is within 1e-3 of fp32 for Keras-initialised
 * and lightly trained towers but not for sharp ones; model.py:31-63 runs fp32).  The weight image
 * then lists, per conv and spatial tap, the planes of Whi and then the planes of Wlo = fp16(W - Whi),
 * each in the plane order above (twice the size of the plain image; a Whi plane serves hi.Whi and
 * lo.Whi while it sits in LDS). */
#define CRL_TRUNK_BITPLANES 1
#define CRL_TRUNK_SPLIT 2
int  crl_trunk_forward_x(void *hip_stream, int filters, int flags, const void *dev_planes,
                         const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                         int n_boards, int n_blocks, const void *dev_head_w_f32,
                         const void *dev_head_b_f32, void *dev_head_out_f32, void *dev_workspace,
                         size_t workspace_bytes);

/* 256 filters with CRL_TRUNK_SPLIT run LAYER-WISE (csrc/tower_layer.hpp: one launch per convolution, a workgroup owns 4
 * boards x all 256 output channels and streams activations and weights; the fused kernel could keep ONE such board resident
 * and streamed the whole weight set for it).  The caller lends the activation images: crl_trunk_workspace_bytes() bytes of
 * device memory (two images of 256 KiB per 4 boards; 0 for every other (filters, flags): dev_workspace may then be NULL),
 * contents undefined before and after a call; workspace_bytes says how much the caller really lent and a call whose batch
 * needs more fails with CRL_ERR_ARG instead of overrunning it (ignored where no workspace is wanted).  n_blocks >= 1 and
 * dev_head_out_f32 are required there (a trunk-only call has no consumer in the product; the other filter counts, whose
 * fused kernels write the fp32 trunk on the way, accept both).  The weight image of
 * that path lists, per convolution, the planes K-CHUNK-major: [conv][in-ch/32][tap][Whi, Wlo][256 rows][4 chunks][8 in]
 * (rows / chunks as above; the stem has 4 chunks of input planes, every other convolution 8). */
size_t crl_trunk_workspace_bytes(int filters, int n_boards, int flags);

/* Hybrid precision (ChessModel(precision="hybrid")).  An evaluation of S1 only chooses the opponent's reply
 * -- argmax of the policy over the legal labels (agentdistributed.py:57-58 -> mctree.py:244-250) -- so S1 runs
 * the single-MFMA trunk first and only the boards whose choice is not safe against that arithmetic's error are
 * evaluated again with CRL_TRUNK_SPLIT; S2 (priors and value: the 1e-3 outputs) always runs CRL_TRUNK_SPLIT.
 * crl_reply_margin lists the unsafe boards from the legal priors (or logits) of the first pass: board b is
 * listed when it has at least two legal moves and log p1 - log p2 of its two best ones (the difference of
 * their logits) is below *dev_log_margin_f32 -- one float in DEVICE memory (not by value: the margin belongs to
 * the weight set, and weights are rewritten in place under captured hipGraphs; a NaN there lists every board).
 * dev_list: int32 [CRL_LIST_HEADER + n_boards]: [0] boards listed by this call, [1] unused, [2..3] one uint64
 * running total over all calls on this buffer (statistics; zero it once), [CRL_LIST_HEADER + k] the boards
 * (unordered).
 * crl_trunk_forward_indexed evaluates exactly the listed boards (rows of dev_bitplanes_u64 / dev_head_out_f32;
 * other rows of dev_head_out_f32 are left as they are): the CRL_TRUNK_BITPLANES | CRL_TRUNK_SPLIT kernels with a
 * grid for n_boards whose surplus workgroups exit at once -- the list never leaves the device, the launch sequence
 * is fixed and captures into a hipGraph.  dev_workspace: as for crl_trunk_forward_x (crl_trunk_workspace_bytes(filters,
 * n_boards, CRL_TRUNK_BITPLANES | CRL_TRUNK_SPLIT); may be the buffer of the full-batch launches).  Stateless. */
#define CRL_LIST_HEADER 4
int  crl_reply_margin(void *hip_stream, const void *dev_priors_f32, const int32_t *dev_counts, int n_boards,
                      const float *dev_log_margin_f32, int rows_are_logits, int32_t *dev_list);
int  crl_trunk_forward_indexed(void *hip_stream, int filters, const void *dev_bitplanes_u64,
                               const void *dev_wtiles_f16x3, const void *dev_bias_f32, int n_boards, int n_blocks,
                               const void *dev_head_w_f32, const void *dev_head_b_f32, void *dev_head_out_f32,
                               const int32_t *dev_list, void *dev_workspace, size_t workspace_bytes);

/* crl_trunk_forward with the input given as plane bitboards (CRL_PLANES_BITS), uint64
 * [n_boards][128]; everything else as above. */
int  crl_trunk_forward_bitplanes(void *hip_stream, int filters, const void *dev_bitplanes_u64,
                                 const void *dev_wtiles_f16, const void *dev_bias_f32, void *dev_out_f32,
                                 int n_boards, int n_blocks, const void *dev_head_w_f32,
                                 const void *dev_head_b_f32, void *dev_head_out_f32);

/* The dense layers behind the head convolutions (model.py:44-48: Dense(1968, softmax); model.py:56-61:
 * Dense(256, relu) -> Dense(1, tanh)) on the [n_boards][192] head activations crl_trunk_forward
 * leaves: one launch per head, fp32-accurate through fp16 MFMAs on (hi, lo)-split operands.
 * dev_policy_wp_f16: the policy kernel W[128 in][1968 out] packed in fragment order
 * [label tile 128][k-step 4][hi|lo][lane 64][8 halves], lane 16 q + r holding
 * W[32 s + 8 q + e][16 tile + r] (labels >= 1968 zero): 1 MiB; dev_policy_bias_f32: [2048], entries
 * >= 1968 set to -1e30.  dev_value_w1p_f16: W1[64 in][256 out] packed the same way
 * [tile 16][k-step 2][hi|lo][lane][8]: 64 KiB; b1 [256]; dev_value_w2b2_f32 [257] = w2 then b2 (in
 * memory, not by value: weights may be rewritten in place under a captured hipGraph).
 * dev_policy_out_f32: [n_boards][1968] probabilities; dev_value_out_f32: [n_boards] or NULL (the
 * value head is skipped: evaluations of S1 only choose the reply).  Stateless (dev_scratch_f32: below). */
int  crl_heads_forward(void *hip_stream, const void *dev_head_act_f32, int n_boards,
                       const void *dev_policy_wp_f16, const void *dev_policy_bias_f32,
                       const void *dev_value_w1p_f16, const void *dev_value_b1_f32,
                       const void *dev_value_w2b2_f32, void *dev_policy_out_f32, void *dev_value_out_f32,
                       void *dev_scratch_f32);

/* crl_heads_forward writing only what the search reads: dev_priors_out_f32 [n_boards][CRL_MAX_MOVES],
 * entry j of row b = softmax(...)[dev_labels[b][j]] for j < dev_counts[b] (the rest of the row is left
 * untouched) -- 4 x count bytes per board instead of 7 872, bit-identical values.  dev_labels uint16
 * [n_boards][CRL_MAX_MOVES], dev_counts int32 [n_boards] (crl_eval_labels).  Stateless. */
int  crl_heads_forward_legal(void *hip_stream, const void *dev_head_act_f32, int n_boards,
                             const void *dev_policy_wp_f16, const void *dev_policy_bias_f32,
                             const void *dev_value_w1p_f16, const void *dev_value_b1_f32,
                             const void *dev_value_w2b2_f32, const uint16_t *dev_labels,
                             const int32_t *dev_counts, void *dev_priors_out_f32, void *dev_value_out_f32,
                             void *dev_scratch_f32);

/* crl_heads_forward_legal without its normalising pass, for batches the sliced form serves
 * (crl_heads_raw_supported(n_boards) == 1; CRL_ERR_ARG otherwise): dev_logits_out_f32 rows receive the
 * LOGITS of the legal moves and dev_stats_out_f32 (float [n_boards][16], required) the slice statistics;
 * the search kernels normalise on read (CRL_POLICY_LEGAL_RAW, crl_set_policy_stats) -- one launch per
 * tower call instead of two, identical priors. */
int  crl_heads_forward_legal_raw(void *hip_stream, const void *dev_head_act_f32, int n_boards,
                                 const void *dev_policy_wp_f16, const void *dev_policy_bias_f32,
                                 const void *dev_value_w1p_f16, const void *dev_value_b1_f32,
                                 const void *dev_value_w2b2_f32, const uint16_t *dev_labels,
                                 const int32_t *dev_counts, void *dev_logits_out_f32, void *dev_value_out_f32,
                                 void *dev_stats_out_f32);
int  crl_heads_raw_supported(int n_boards);

/* dev_scratch_f32 of the two calls above: float [n_boards][16] or NULL.  With it, batches of at most
 * crl_heads_set_sliced_max boards (default 2048: measured 16 vs 17 us there, 9 vs 17 us at 512) run as label slices x board blocks (8 slices of 256
 * labels, the value head as a ninth "slice": 9 x n_boards/16 workgroups instead of n_boards/16)
 * followed by a normalising pass that reads the slice statistics left in the scratch -- a batch of
 * 512 boards otherwise pulls the whole 1-MiB policy kernel through each of 32 CUs.  Same results for
 * the two calls (bit for bit); the last bits differ from the one-pass form (summation order).  Larger batches run ONE launch: the
 * policy head's workgroups and, behind them, the value head's.
 * PROCESS-GLOBAL tuning state (like crl_trunk_set_small_batch below): it applies to every context and
 * stream of the process; meant to be set once at start-up, not per context. */
int  crl_heads_set_sliced_max(int boards);

/* Batches of at most 512 boards (256 at 256 filters) give at most half of the 256 CUs a
 * workgroup; they run the same kernels with half the boards per workgroup and twice the
 * workgroups (identical trunk bits).  enabled = 0 turns that off process-wide (default 1). */
int  crl_trunk_set_small_batch(int enabled);

/* Which kernel crl_trunk_forward_x launches for this filter count, batch and flags, written as
 * rocprofv3 prints it without namespace and argument list ("k_trunk_x16<128, 4, 1, 0, 1, 0, 0, 0>"):
 * measurement tools look their profiles up by it instead of restating the dispatch rule.  No
 * reference counterpart (model.py:31-63 builds one Keras graph). */
int  crl_trunk_kernel_name(int filters, int n_boards, int flags, char *buf, int buf_len);

/* ---- measurement -------------------------------------------------------------------------------
 * crl_stamp enqueues a one-thread kernel that appends (id, wall clock of the device) to a ring in device memory:
 * dev_ring is uint64 [2 + 2 * capacity], [0] = stamps written so far (zero it once), [1] unused, then pairs
 * (id, clock) at entry (count % capacity).  It captures into a hipGraph like any kernel, so bench.py gets the time of
 * every phase of a step and of every trunk launch from the REPLAYED graph (consecutive stamps telescope to the step)
 * instead of from eager launches beside it.  crl_stamp_clock_khz: the rate of that clock
 * (hipDeviceAttributeWallClockRate; 100 000 on MI355X), or a negative crl_status.  No reference counterpart. */
int  crl_stamp(void *hip_stream, uint64_t *dev_ring, uint32_t capacity, uint32_t id);
int  crl_stamp_clock_khz(int device);

/* ---- training step (model.py:83-99 fit_generator) --------------------
 * The reference's Conv2D layers (model.py:33-34,113-118) train through TensorFlow; here a 3x3 'same'
 * convolution on the 8x8 board is the GEMM [B*64, 9*C] x [9*C, Cout] over NHWC activations, and these
 * two kernels build its patch matrix and the adjoint (gradient w.r.t. the activations).  fp32, NHWC
 * x [n_boards][8][8][C], cols [n_boards*64][9][C] with tap = ky*3+kx; C % 4 == 0.  Stateless. */
int  crl_im2col3x3_f32(void *hip_stream, const void *dev_x_f32, void *dev_cols_f32, int n_boards,
                       int channels);
int  crl_col2im3x3_f32(void *hip_stream, const void *dev_gcols_f32, void *dev_gx_f32, int n_boards,
                       int channels);

#ifdef __cplusplus
}
#endif
#endif /* CHESSRL_HIP_H */
