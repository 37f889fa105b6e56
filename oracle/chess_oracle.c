/*
 * oracle/chess_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the chess-rules half of the reference's self-play
 * hot path: Game.get_legal_moves / Game.move / Game.get_result / Game.get_copy
 * (/root/reference/src/chessrl/game.py:28-57, 79-80, 92-109).  Those reference
 * functions delegate all arithmetic to the third-party package
 * python-chess==0.28.3 (/root/reference/requirements.txt:9), whose source is
 * NOT under /root/reference and is not installed in this image.  This file
 * therefore restates python-chess's *published algorithm* (chess/__init__.py of
 * that release: generate_legal_moves, _generate_evasions,
 * generate_pseudo_legal_moves, generate_castling_moves, push, is_game_over,
 * result, can_claim_fifty_moves, is_insufficient_material, is_repetition,
 * _transposition_key) from memory.
 *
 * PARITY STATUS: the legal-move SET is pinned by public perft known answers
 * (tests/test_oracle_chess.py); the move ORDER is pinned only by the published
 * start-position listing of python-chess's README and by the algorithm
 * restated here -- "parity unpinned" against python-chess itself.
 *
 * Deliberately a different algorithm from the GPU kernels: mailbox board,
 * ray walking for attacks, legality by make-move + king-attacked test (no pin
 * masks, no bitboard attack tables).  Only the ORDER of emission follows
 * python-chess, because order is part of the contract (mctree.py:55-56 pops
 * the LAST legal move first; np.argmax tie-breaks by position).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- shared POD layouts (must match include/chessrl_hip.h) ---------------- */
typedef struct {
    uint64_t bb[6];   /* pawns, knights, bishops, rooks, queens, kings (both colours) */
    uint64_t white;   /* squares occupied by white                                     */
    uint32_t state;   /* bit0 turn(1=white) | bits1-4 castling WK,WQ,BK,BQ |
                         bits5-11 ep square (64 = none) | bits12-19 halfmove clock |
                         bit20 "a legal en-passant capture exists"                    */
    uint32_t pad;
} oc_board;

#define ST_TURN(s)   ((s) & 1u)
#define ST_CASTLE(s) (((s) >> 1) & 15u)
#define ST_EP(s)     (((s) >> 5) & 127u)
#define ST_CLOCK(s)  (((s) >> 12) & 255u)
#define ST_EPLEGAL(s) (((s) >> 20) & 1u)
#define MK_STATE(turn, castle, ep, clock, epl) \
    ((uint32_t)(turn) | ((uint32_t)(castle) << 1) | ((uint32_t)(ep) << 5) | \
     ((uint32_t)(clock) << 12) | ((uint32_t)(epl) << 20))

enum { EMPTY = 0, PAWN = 1, KNIGHT = 2, BISHOP = 3, ROOK = 4, QUEEN = 5, KING = 6 };
enum { CR_WK = 1, CR_WQ = 2, CR_BK = 4, CR_BQ = 8 };
#define NO_EP 64
#define RESULT_NONE 2

/* mailbox position: piece[sq] = type, colour[sq] = 1 white / 0 black */
typedef struct {
    int8_t piece[64];
    int8_t colour[64];
    int turn, castle, ep, clock;
} mbx;

static void to_mbx(const oc_board *b, mbx *m)
{
    memset(m, 0, sizeof *m);
    for (int t = 0; t < 6; t++)
        for (int sq = 0; sq < 64; sq++)
            if ((b->bb[t] >> sq) & 1) {
                m->piece[sq] = (int8_t)(t + 1);
                m->colour[sq] = (int8_t)((b->white >> sq) & 1);
            }
    m->turn = ST_TURN(b->state);
    m->castle = ST_CASTLE(b->state);
    m->ep = ST_EP(b->state);
    m->clock = ST_CLOCK(b->state);
}

static int has_legal_ep(const mbx *m);

static void from_mbx(const mbx *m, oc_board *b)
{
    memset(b, 0, sizeof *b);
    for (int sq = 0; sq < 64; sq++)
        if (m->piece[sq]) {
            b->bb[m->piece[sq] - 1] |= 1ull << sq;
            if (m->colour[sq]) b->white |= 1ull << sq;
        }
    b->state = MK_STATE(m->turn, m->castle, m->ep, m->clock, has_legal_ep(m));
}

/* ---- attacks by ray walking ---------------------------------------------- */
static const int KN_DF[8] = { 1, 2, 2, 1, -1, -2, -2, -1 };
static const int KN_DR[8] = { 2, 1, -1, -2, -2, -1, 1, 2 };
static const int K_DF[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };
static const int K_DR[8] = { 1, 0, -1, 1, -1, 1, 0, -1 };
static const int B_DF[4] = { 1, 1, -1, -1 };
static const int B_DR[4] = { 1, -1, 1, -1 };
static const int R_DF[4] = { 1, -1, 0, 0 };
static const int R_DR[4] = { 0, 0, 1, -1 };

static int on(int f, int r) { return f >= 0 && f < 8 && r >= 0 && r < 8; }

/* is `sq` attacked by side `by` (1 white / 0 black) on mailbox m? */
static int attacked(const mbx *m, int sq, int by)
{
    int f = sq & 7, r = sq >> 3;
    /* pawns: a white pawn on (f±1, r-1) attacks sq */
    int pr = by ? r - 1 : r + 1;
    for (int df = -1; df <= 1; df += 2)
        if (on(f + df, pr)) {
            int s = pr * 8 + f + df;
            if (m->piece[s] == PAWN && m->colour[s] == by) return 1;
        }
    for (int i = 0; i < 8; i++) {
        if (on(f + KN_DF[i], r + KN_DR[i])) {
            int s = (r + KN_DR[i]) * 8 + f + KN_DF[i];
            if (m->piece[s] == KNIGHT && m->colour[s] == by) return 1;
        }
        if (on(f + K_DF[i], r + K_DR[i])) {
            int s = (r + K_DR[i]) * 8 + f + K_DF[i];
            if (m->piece[s] == KING && m->colour[s] == by) return 1;
        }
    }
    for (int d = 0; d < 4; d++) {
        int cf = f + B_DF[d], cr = r + B_DR[d];
        while (on(cf, cr)) {
            int s = cr * 8 + cf;
            if (m->piece[s]) {
                if (m->colour[s] == by && (m->piece[s] == BISHOP || m->piece[s] == QUEEN)) return 1;
                break;
            }
            cf += B_DF[d]; cr += B_DR[d];
        }
        cf = f + R_DF[d]; cr = r + R_DR[d];
        while (on(cf, cr)) {
            int s = cr * 8 + cf;
            if (m->piece[s]) {
                if (m->colour[s] == by && (m->piece[s] == ROOK || m->piece[s] == QUEEN)) return 1;
                break;
            }
            cf += R_DF[d]; cr += R_DR[d];
        }
    }
    return 0;
}

static int king_sq(const mbx *m, int side)
{
    /* python-chess uses msb(kings & occupied_co[side]); with one king it is that king */
    for (int sq = 63; sq >= 0; sq--)
        if (m->piece[sq] == KING && m->colour[sq] == side) return sq;
    return -1;
}

/* pseudo-attack target mask of the non-pawn piece on `sq` (excludes own pieces) */
static uint64_t piece_targets(const mbx *m, int sq)
{
    uint64_t t = 0;
    int f = sq & 7, r = sq >> 3, me = m->colour[sq], p = m->piece[sq];
    if (p == KNIGHT || p == KING) {
        const int *df = p == KNIGHT ? KN_DF : K_DF, *dr = p == KNIGHT ? KN_DR : K_DR;
        for (int i = 0; i < 8; i++)
            if (on(f + df[i], r + dr[i])) {
                int s = (r + dr[i]) * 8 + f + df[i];
                if (!m->piece[s] || m->colour[s] != me) t |= 1ull << s;
            }
        return t;
    }
    for (int d = 0; d < 4; d++) {
        if (p == BISHOP || p == QUEEN) {
            int cf = f + B_DF[d], cr = r + B_DR[d];
            while (on(cf, cr)) {
                int s = cr * 8 + cf;
                if (m->piece[s]) { if (m->colour[s] != me) t |= 1ull << s; break; }
                t |= 1ull << s; cf += B_DF[d]; cr += B_DR[d];
            }
        }
        if (p == ROOK || p == QUEEN) {
            int cf = f + R_DF[d], cr = r + R_DR[d];
            while (on(cf, cr)) {
                int s = cr * 8 + cf;
                if (m->piece[s]) { if (m->colour[s] != me) t |= 1ull << s; break; }
                t |= 1ull << s; cf += R_DF[d]; cr += R_DR[d];
            }
        }
    }
    return t;
}

#define MV(from, to, promo) ((uint16_t)((from) | ((to) << 6) | ((promo) << 12)))
#define MV_FROM(m) ((m) & 63)
#define MV_TO(m) (((m) >> 6) & 63)
#define MV_PROMO(m) (((m) >> 12) & 7)

/* apply a (pseudo-legal) move to a mailbox; mirrors python-chess Board.push */
static void make(mbx *m, uint16_t mv)
{
    int from = MV_FROM(mv), to = MV_TO(mv), promo = MV_PROMO(mv);
    int p = m->piece[from], me = m->colour[from];
    int captured = m->piece[to];
    int ep = m->ep;
    m->ep = NO_EP;
    /* is_zeroing: pawn move or capture */
    if (p == PAWN || captured) m->clock = 0; else if (m->clock < 255) m->clock++;
    /* castling rights: touched rook squares / king moves */
    if (from == 7 || to == 7) m->castle &= ~CR_WK;
    if (from == 0 || to == 0) m->castle &= ~CR_WQ;
    if (from == 63 || to == 63) m->castle &= ~CR_BK;
    if (from == 56 || to == 56) m->castle &= ~CR_BQ;
    if (p == KING) m->castle &= me ? ~(CR_WK | CR_WQ) : ~(CR_BK | CR_BQ);

    m->piece[from] = EMPTY;
    if (p == PAWN) {
        int diff = to - from;
        if (diff == 16 && (from >> 3) == 1) m->ep = from + 8;
        else if (diff == -16 && (from >> 3) == 6) m->ep = from - 8;
        else if (to == ep && (diff == 7 || diff == 9 || diff == -7 || diff == -9) && !captured) {
            int cap = me ? to - 8 : to + 8;
            m->piece[cap] = EMPTY;
        }
    }
    if (p == KING && (to - from == 2 || from - to == 2)) {
        /* castling given as king move e1g1/e1c1: also move the rook */
        int rf = to > from ? from + 3 : from - 4, rt = to > from ? from + 1 : from - 1;
        m->piece[rt] = ROOK; m->colour[rt] = (int8_t)me;
        m->piece[rf] = EMPTY;
    }
    m->piece[to] = (int8_t)(promo ? promo : p);
    m->colour[to] = (int8_t)me;
    m->turn ^= 1;
}

static int legal_after(const mbx *m, uint16_t mv)
{
    mbx c = *m;
    int me = m->turn;
    make(&c, mv);
    int k = king_sq(&c, me);
    return k < 0 || !attacked(&c, k, !me);
}

/* pseudo-legal en-passant captures, capturers high -> low */
static int gen_ep(const mbx *m, uint16_t *out)
{
    int n = 0;
    if (m->ep == NO_EP || m->piece[m->ep]) return 0;
    int me = m->turn, er = m->ep >> 3, ef = m->ep & 7;
    int cr = me ? 4 : 3;                 /* rank index of capturing pawns */
    if (er != (me ? 5 : 2)) return 0;    /* BB_PAWN_ATTACKS[!turn][ep] & BB_RANKS[4 or 3] */
    for (int cf = ef + 1; cf >= ef - 1; cf -= 2)
        if (cf >= 0 && cf < 8) {
            int s = cr * 8 + cf;
            if (m->piece[s] == PAWN && m->colour[s] == me) out[n++] = MV(s, m->ep, 0);
        }
    return n;
}

static int has_legal_ep(const mbx *m)
{
    uint16_t ep[2];
    int n = gen_ep(m, ep);
    for (int i = 0; i < n; i++) if (legal_after(m, ep[i])) return 1;
    return 0;
}

/*
 * Legal moves in python-chess 0.28.3 generation order.
 *   not in check: non-pawn pieces (from high->low, to high->low); castling
 *   (h-side then a-side); pawn captures (from high->low, to high->low,
 *   promotions Q,R,B,N); single pushes by to-square high->low; double pushes;
 *   en passant.
 *   in check (_generate_evasions): king steps first (to high->low); then, for
 *   a single checker, the same sequence for non-king pieces; then en passant.
 */
static int gen_legal(const mbx *m, uint16_t *out)
{
    int n = 0, me = m->turn;
    int ksq = king_sq(m, me);
    int in_check = ksq >= 0 && attacked(m, ksq, !me);
    uint16_t cand[256];
    int nc = 0;

    if (in_check) {
        uint64_t t = piece_targets(m, ksq);
        for (int to = 63; to >= 0; to--)
            if ((t >> to) & 1) cand[nc++] = MV(ksq, to, 0);
    }
    for (int from = 63; from >= 0; from--) {
        if (!m->piece[from] || m->colour[from] != me || m->piece[from] == PAWN) continue;
        if (in_check && m->piece[from] == KING) continue;
        uint64_t t = piece_targets(m, from);
        for (int to = 63; to >= 0; to--)
            if ((t >> to) & 1) cand[nc++] = MV(from, to, 0);
    }
    if (!in_check && ksq == (me ? 4 : 60)) {
        /* castling: candidate rooks high -> low => king side first */
        int base = me ? 0 : 56;
        if ((m->castle & (me ? CR_WK : CR_BK)) && m->piece[base + 7] == ROOK &&
            m->colour[base + 7] == me && !m->piece[base + 5] && !m->piece[base + 6] &&
            !attacked(m, base + 5, !me) && !attacked(m, base + 6, !me))
            cand[nc++] = MV(ksq, base + 6, 0);
        if ((m->castle & (me ? CR_WQ : CR_BQ)) && m->piece[base + 0] == ROOK &&
            m->colour[base + 0] == me && !m->piece[base + 1] && !m->piece[base + 2] &&
            !m->piece[base + 3] && !attacked(m, base + 3, !me) && !attacked(m, base + 2, !me))
            cand[nc++] = MV(ksq, base + 2, 0);
    }
    /* pawn captures */
    for (int from = 63; from >= 0; from--) {
        if (m->piece[from] != PAWN || m->colour[from] != me) continue;
        int f = from & 7, r = from >> 3, tr = me ? r + 1 : r - 1;
        if (tr < 0 || tr > 7) continue;
        for (int tf = f + 1; tf >= f - 1; tf -= 2) {   /* higher square first */
            if (tf < 0 || tf > 7) continue;
            int to = tr * 8 + tf;
            if (!m->piece[to] || m->colour[to] == me) continue;
            if (tr == 0 || tr == 7) {
                cand[nc++] = MV(from, to, QUEEN); cand[nc++] = MV(from, to, ROOK);
                cand[nc++] = MV(from, to, BISHOP); cand[nc++] = MV(from, to, KNIGHT);
            } else cand[nc++] = MV(from, to, 0);
        }
    }
    /* single pushes ordered by to-square high -> low */
    for (int to = 63; to >= 0; to--) {
        int from = me ? to - 8 : to + 8;
        if (from < 0 || from > 63 || m->piece[to]) continue;
        if (m->piece[from] != PAWN || m->colour[from] != me) continue;
        if ((to >> 3) == 0 || (to >> 3) == 7) {
            cand[nc++] = MV(from, to, QUEEN); cand[nc++] = MV(from, to, ROOK);
            cand[nc++] = MV(from, to, BISHOP); cand[nc++] = MV(from, to, KNIGHT);
        } else cand[nc++] = MV(from, to, 0);
    }
    /* double pushes ordered by to-square high -> low */
    for (int to = 63; to >= 0; to--) {
        if ((to >> 3) != (me ? 3 : 4) || m->piece[to]) continue;
        int mid = me ? to - 8 : to + 8, from = me ? to - 16 : to + 16;
        if (m->piece[mid]) continue;
        if (m->piece[from] != PAWN || m->colour[from] != me) continue;
        cand[nc++] = MV(from, to, 0);
    }
    nc += gen_ep(m, cand + nc);

    for (int i = 0; i < nc; i++)
        if (legal_after(m, cand[i])) out[n++] = cand[i];
    return n;
}

/* ---- insufficient material (python-chess has_insufficient_material) ------- */
static int popc(uint64_t x) { return __builtin_popcountll(x); }

static int side_insufficient(const oc_board *b, int colour)
{
    const uint64_t DARK = 0xAA55AA55AA55AA55ull, LIGHT = 0x55AA55AA55AA55AAull;
    uint64_t occ = b->bb[0] | b->bb[1] | b->bb[2] | b->bb[3] | b->bb[4] | b->bb[5];
    uint64_t own = colour ? b->white : (occ & ~b->white), opp = occ & ~own;
    if (own & (b->bb[0] | b->bb[3] | b->bb[4])) return 0;
    if (own & b->bb[1])
        return popc(own) <= 2 && !(opp & ~b->bb[5] & ~b->bb[4]);
    if (own & b->bb[2]) {
        int same = !(b->bb[2] & DARK) || !(b->bb[2] & LIGHT);
        return same && !b->bb[0] && !b->bb[1];
    }
    return 1;
}

/* ---- game object: full position + move history ---------------------------- */
typedef struct {
    int ply, cap;
    oc_board *pos;      /* pos[0..ply], pos[ply] = current */
    uint16_t *moves;    /* moves[0..ply-1] */
} og;

static const oc_board START = {
    { 0x00FF00000000FF00ull, 0x4200000000000042ull, 0x2400000000000024ull,
      0x8100000000000081ull, 0x0800000000000008ull, 0x1000000000000010ull },
    0x000000000000FFFFull,
    MK_STATE(1, 15, NO_EP, 0, 0), 0
};

og *og_new(void)
{
    og *g = (og *)calloc(1, sizeof *g);
    g->cap = 64;
    g->pos = (oc_board *)malloc(sizeof(oc_board) * (g->cap + 1));
    g->moves = (uint16_t *)malloc(sizeof(uint16_t) * g->cap);
    g->pos[0] = START;
    return g;
}

/* start a game from an arbitrary position with empty history (ep-legal bit recomputed) */
og *og_from_board(const oc_board *b)
{
    og *g = og_new();
    mbx m; to_mbx(b, &m);
    from_mbx(&m, &g->pos[0]);
    return g;
}

void og_free(og *g) { if (g) { free(g->pos); free(g->moves); free(g); } }

og *og_copy(const og *s)
{
    og *g = (og *)calloc(1, sizeof *g);
    g->ply = s->ply; g->cap = s->cap;
    g->pos = (oc_board *)malloc(sizeof(oc_board) * (g->cap + 1));
    g->moves = (uint16_t *)malloc(sizeof(uint16_t) * g->cap);
    memcpy(g->pos, s->pos, sizeof(oc_board) * (s->ply + 1));
    memcpy(g->moves, s->moves, sizeof(uint16_t) * s->ply);
    return g;
}

int og_ply(const og *g) { return g->ply; }
int og_turn(const og *g) { return ST_TURN(g->pos[g->ply].state); }
void og_board(const og *g, int back, oc_board *out) { *out = g->pos[g->ply - back]; }
uint16_t og_move_at(const og *g, int i) { return g->moves[i]; }

int og_legal_moves(const og *g, uint16_t *out)
{
    mbx m; to_mbx(&g->pos[g->ply], &m);
    return gen_legal(&m, out);
}

/* Game.move (game.py:28-41): apply iff in the legal list; returns 1/0 */
int og_push(og *g, uint16_t mv)
{
    uint16_t lm[256];
    mbx m; to_mbx(&g->pos[g->ply], &m);
    int n = gen_legal(&m, lm), ok = 0;
    for (int i = 0; i < n; i++) if (lm[i] == mv) ok = 1;
    if (!ok) return 0;
    if (g->ply == g->cap) {
        g->cap *= 2;
        g->pos = (oc_board *)realloc(g->pos, sizeof(oc_board) * (g->cap + 1));
        g->moves = (uint16_t *)realloc(g->moves, sizeof(uint16_t) * g->cap);
    }
    make(&m, mv);
    g->moves[g->ply] = mv;
    g->ply++;
    from_mbx(&m, &g->pos[g->ply]);
    return 1;
}

static int same_key(const oc_board *a, const oc_board *b)
{
    /* python-chess _transposition_key: piece sets, occupancies, turn, clean
       castling rights, ep square only if a legal ep capture exists */
    if (memcmp(a->bb, b->bb, sizeof a->bb) || a->white != b->white) return 0;
    if (ST_TURN(a->state) != ST_TURN(b->state)) return 0;
    if (ST_CASTLE(a->state) != ST_CASTLE(b->state)) return 0;
    unsigned ea = ST_EPLEGAL(a->state) ? ST_EP(a->state) : NO_EP;
    unsigned eb = ST_EPLEGAL(b->state) ? ST_EP(b->state) : NO_EP;
    return ea == eb;
}

/* number of occurrences of the current position in the whole game (>= 1) */
int og_repetitions(const og *g)
{
    int c = 1;
    for (int i = g->ply - 1; i >= 0; i--)
        if (same_key(&g->pos[i], &g->pos[g->ply])) c++;
    return c;
}

int og_in_check(const og *g)
{
    mbx m; to_mbx(&g->pos[g->ply], &m);
    int k = king_sq(&m, m.turn);
    return k >= 0 && attacked(&m, k, !m.turn);
}

int og_insufficient(const og *g)
{
    return side_insufficient(&g->pos[g->ply], 1) && side_insufficient(&g->pos[g->ply], 0);
}

/*
 * Game.get_result (game.py:92-109): 0 if can_claim_fifty_moves(); elif
 * is_game_over() -> result() mapped to 1 / -1 / 0; else None (2 here).
 */
int og_result(const og *g)
{
    uint16_t lm[256];
    mbx m; to_mbx(&g->pos[g->ply], &m);
    int n = gen_legal(&m, lm);
    int clock = m.clock;
    if (clock >= 100 && n > 0) return 0;                    /* can_claim_fifty_moves */
    int over = (clock >= 150 && n > 0) || og_insufficient(g) || n == 0 ||
               og_repetitions(g) >= 5;
    if (!over) return RESULT_NONE;
    if (n == 0 && og_in_check(g)) return m.turn ? -1 : 1;   /* checkmate: side to move lost */
    return 0;
}

/* perft for the known-answer tests */
static uint64_t perft_mbx(const mbx *m, int depth)
{
    uint16_t lm[256];
    int n = gen_legal(m, lm);
    if (depth == 1) return (uint64_t)n;
    uint64_t t = 0;
    for (int i = 0; i < n; i++) {
        mbx c = *m; make(&c, lm[i]);
        t += perft_mbx(&c, depth - 1);
    }
    return t;
}

uint64_t og_perft(const og *g, int depth)
{
    mbx m; to_mbx(&g->pos[g->ply], &m);
    return depth <= 0 ? 1 : perft_mbx(&m, depth);
}
