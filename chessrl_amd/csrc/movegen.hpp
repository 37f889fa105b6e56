// movegen.hpp -- one wavefront generates the legal moves of one position.
//
// Replaces Game.get_legal_moves (/root/reference/src/chessrl/game.py:43-57), i.e.
// python-chess 0.28.3 Board.generate_legal_moves, INCLUDING its emission order
// (SURVEY.md Appendix C), because order is part of the reference's contract:
// mctree.py:55-56 expands the LAST legal move first and every np.argmax breaks
// ties by list position.
//
// Mapping: lane l <-> square l (64 lanes = 64 squares).  Each lane computes the
// legal target set of the piece on its square from bitboards held in registers
// (sliders by hyperbola quintessence, pins/check masks from wave-uniform
// arithmetic, the opponent's attack map as one __ballot).  Emission order is
// produced without sorting: python-chess scans from-squares high->low, so a
// move's slot is [category base] + [suffix sum of the counts of higher lanes]
// + [rank of the target inside the lane]; the five category counts are packed
// into one 32-bit word and suffix-scanned with 6 wave shuffles.
//
//   not in check: A non-pawn pieces (king included, by from-square)
//                 C castling (king side, then queen side)
//                 P pawn captures (promotions Q,R,B,N)
//                 S single pushes   D double pushes   E en passant
//   in check    : K king steps first, then A (without king) P S D E restricted
//                 to capturing/blocking a single checker.
#pragma once
#include "board.hpp"

namespace crl {

struct MoveGenInfo {
    int n;            // number of legal moves written
    bool in_check;
    bool ep_legal;    // python-chess has_legal_en_passant()
};

__device__ inline u64 between_sq(int a, int c)
{
    u64 cb = bit(c), m;
    if (rank_mask(a) & cb) m = rank_mask(a);
    else if (file_mask(a) & cb) m = file_mask(a);
    else if (diag_mask(a) & cb) m = diag_mask(a);
    else if (anti_mask(a) & cb) m = anti_mask(a);
    else return 0;
    return line_attacks(a, cb, m) & line_attacks(c, bit(a), m);
}

// inclusive sum over lanes >= my lane (6 shuffle steps)
__device__ inline u32 wave_suffix_sum(u32 v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_down(v, d);
        v += (lane + d < 64) ? t : 0u;
    }
    return v;
}

__device__ inline u16 mk_move(int from, int to, int promo)
{
    return (u16)(from | (to << 6) | (promo << 12));
}

// emit the targets of one lane high->low, promotions expanded Q,R,B,N
__device__ inline void emit_targets(u16 *out, int off, int from, u64 t, bool promo)
{
    while (t) {
        int to = msb(t);
        t &= ~bit(to);
        if (promo) {
            if (off + 3 < 256) {
                out[off] = mk_move(from, to, 5); out[off + 1] = mk_move(from, to, 4);
                out[off + 2] = mk_move(from, to, 3); out[off + 3] = mk_move(from, to, 2);
            }
            off += 4;
        } else {
            if (off < 256) out[off] = mk_move(from, to, 0);
            off += 1;
        }
    }
}

// Wave-cooperative legal move generation.  `out` (LDS or global, 256 entries) receives
// the moves in python-chess order; pass nullptr to only count.  All 64 lanes must call.
// The caller must synchronise before other lanes read `out`.
__device__ inline MoveGenInfo wave_movegen(const Board &b, int lane, u16 *out)
{
    const bool white = st_turn(b.state);
    const u64 occ = occupied(b);
    const u64 own = white ? b.white : (occ & ~b.white);
    const u64 opp = occ & ~own;
    const u64 kbb = b.bb[KING] & own;
    const int ksq = kbb ? msb(kbb) : 0;
    const u64 lb = bit(lane);

    // opponent's attack map with our king lifted off the board: one ballot
    const u64 danger = kbb ? __ballot(attackers_to(b, lane, occ & ~kbb, !white) != 0) : 0ull;
    const u64 checkers = kbb ? attackers_to(b, ksq, occ, !white) : 0ull;
    const int nchk = popc(checkers);
    const bool in_check = nchk > 0;
    u64 check_mask = ~0ull;
    if (nchk == 1) check_mask = checkers | between_sq(ksq, lsb(checkers));
    else if (nchk > 1) check_mask = 0;

    const bool mine = (own & lb) != 0;
    const int pt = piece_at(b, lane);

    // absolute pin of the piece on this lane's square
    u64 pin_mask = ~0ull;
    if (mine && kbb && lane != ksq) {
        u64 line = 0, slid = 0;
        if (rank_mask(ksq) & lb) { line = rank_mask(ksq); slid = b.bb[ROOK] | b.bb[QUEEN]; }
        else if (file_mask(ksq) & lb) { line = file_mask(ksq); slid = b.bb[ROOK] | b.bb[QUEEN]; }
        else if (diag_mask(ksq) & lb) { line = diag_mask(ksq); slid = b.bb[BISHOP] | b.bb[QUEEN]; }
        else if (anti_mask(ksq) & lb) { line = anti_mask(ksq); slid = b.bb[BISHOP] | b.bb[QUEEN]; }
        if (line) {
            u64 a1 = line_attacks(ksq, occ, line);
            if (a1 & lb) {
                u64 a2 = line_attacks(ksq, occ & ~lb, line);
                if ((a2 & ~a1) & opp & slid) pin_mask = line;
            }
        }
    }

    // ---- per-lane target sets ---------------------------------------------------
    u64 tA = 0, tP = 0, tS = 0, tD = 0, tE = 0;
    bool promo = false;
    const u64 king_tgt = kbb ? (king_attacks(ksq) & ~own & ~danger) : 0ull;   // uniform
    if (mine && pt != PAWN) {
        if (pt == KING) tA = in_check ? 0ull : king_tgt;
        else {
            u64 att = pt == KNIGHT ? knight_attacks(lane)
                    : pt == BISHOP ? bishop_attacks(lane, occ)
                    : pt == ROOK   ? rook_attacks(lane, occ)
                                   : (bishop_attacks(lane, occ) | rook_attacks(lane, occ));
            tA = att & ~own & pin_mask & check_mask;
        }
    }
    if (mine && pt == PAWN) {
        const u64 legal_to = pin_mask & check_mask;
        const u64 patt = pawn_attacks(lane, white);
        tP = patt & opp & legal_to;
        const int r = lane >> 3;
        const int to1 = white ? lane + 8 : lane - 8;
        if (to1 >= 0 && to1 < 64) {
            promo = (to1 >> 3) == (white ? 7 : 0);
            const bool free1 = !(occ & bit(to1));
            if (free1) tS = bit(to1) & legal_to;
            if (free1 && r == (white ? 1 : 6)) {
                const int to2 = white ? lane + 16 : lane - 16;
                if (!(occ & bit(to2))) tD = bit(to2) & legal_to;
            }
        }
        const int ep = (int)st_ep(b.state);
        if (ep != NO_EP && (patt & bit(ep)) && !(occ & bit(ep)) && r == (white ? 4 : 3)) {
            // play it on the occupancy and look for any attacker of our king
            const int cap = white ? ep - 8 : ep + 8;
            const u64 occ2 = (occ & ~lb & ~bit(cap)) | bit(ep);
            if (!kbb || !attackers_to(b, ksq, occ2, !white)) tE = bit(ep);
        }
    }

    // ---- counts -> slots ---------------------------------------------------------
    const u32 cA = popc(tA), cP = popc(tP) * (promo ? 4 : 1), cS = tS ? (promo ? 4u : 1u) : 0u;
    const u32 cD = tD ? 1u : 0u, cE = tE ? 1u : 0u;
    const u32 packed = cA | (cP << 10) | (cS << 20) | (cD << 26) | (cE << 30);
    const u32 incl = wave_suffix_sum(packed, lane);
    const u32 tot = __shfl(incl, 0);
    const u32 excl = incl - packed;
    const int totA = tot & 1023, totP = (tot >> 10) & 1023, totS = (tot >> 20) & 63;
    const int totD = (tot >> 26) & 15, totE = (tot >> 30) & 3;
    const int nK = in_check ? popc(king_tgt) : 0;

    bool cs_k = false, cs_q = false;
    if (!in_check && kbb && ksq == (white ? 4 : 60)) {
        const int base = white ? 0 : 56;
        const u32 cr = st_castle(b.state);
        const u64 rooks = b.bb[ROOK] & own;
        const u64 f = bit(base + 5), g = bit(base + 6);
        const u64 bq = bit(base + 1), c = bit(base + 2), d = bit(base + 3);
        cs_k = (cr & (white ? CR_WK : CR_BK)) && (rooks & bit(base + 7)) &&
               !(occ & (f | g)) && !(danger & (f | g));
        cs_q = (cr & (white ? CR_WQ : CR_BQ)) && (rooks & bit(base + 0)) &&
               !(occ & (bq | c | d)) && !(danger & (c | d));
    }
    const int nC = (cs_k ? 1 : 0) + (cs_q ? 1 : 0);

    const int baseA = nK, baseC = baseA + totA, baseP = baseC + nC, baseS = baseP + totP;
    const int baseD = baseS + totS, baseE = baseD + totD;
    const int total = baseE + totE;

    if (out) {
        if (in_check && lane == ksq) emit_targets(out, 0, lane, king_tgt, false);
        if (cA) emit_targets(out, baseA + (int)(excl & 1023), lane, tA, false);
        if (lane == 0) {
            int o = baseC;
            if (cs_k && o < 256) out[o++] = mk_move(ksq, ksq + 2, 0);
            if (cs_q && o < 256) out[o++] = mk_move(ksq, ksq - 2, 0);
        }
        if (cP) emit_targets(out, baseP + (int)((excl >> 10) & 1023), lane, tP, promo);
        if (cS) emit_targets(out, baseS + (int)((excl >> 20) & 63), lane, tS, promo);
        if (cD) emit_targets(out, baseD + (int)((excl >> 26) & 15), lane, tD, false);
        if (cE) emit_targets(out, baseE + (int)((excl >> 30) & 3), lane, tE, false);
    }
    MoveGenInfo info;
    info.n = total > 256 ? 256 : total;
    info.in_check = in_check;
    info.ep_legal = totE > 0;
    return info;
}

}  // namespace crl
