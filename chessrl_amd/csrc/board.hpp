// board.hpp -- bitboard primitives for gfx950 (device side).
//
// Rules semantics follow python-chess 0.28.3, the third-party package behind
// the reference's Game wrapper (/root/reference/src/chessrl/game.py:28-109):
// Board.push (castling rights, ep square on every double push, halfmove
// clock), has_insufficient_material, _transposition_key.  Squares a1=0..h8=63.
//
// Everything in this file is wave-UNIFORM arithmetic unless a parameter is
// named `lane`/`sq`: all 64 lanes compute the same value (the compiler is free
// to keep it in SGPRs).  Sliding attacks use hyperbola quintessence with the
// hardware bit-reverse (v_bfrev_b32 / s_brev_b64), no lookup tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;

namespace crl {

struct __attribute__((aligned(16))) Board {
    u64 bb[6];    // P N B R Q K (both colours)
    u64 white;
    u32 state;
    u32 pad;
};
static_assert(sizeof(Board) == 64, "Board must be 64 bytes");

enum { PAWN = 0, KNIGHT = 1, BISHOP = 2, ROOK = 3, QUEEN = 4, KING = 5 };
enum { CR_WK = 1, CR_WQ = 2, CR_BK = 4, CR_BQ = 8 };
constexpr int NO_EP = 64;
constexpr int RESULT_NONE = 2;
constexpr u16 NO_MOVE = 0xFFFF;

__host__ __device__ inline u32 st_turn(u32 s) { return s & 1u; }
__host__ __device__ inline u32 st_castle(u32 s) { return (s >> 1) & 15u; }
__host__ __device__ inline u32 st_ep(u32 s) { return (s >> 5) & 127u; }
__host__ __device__ inline u32 st_clock(u32 s) { return (s >> 12) & 255u; }
__host__ __device__ inline u32 st_eplegal(u32 s) { return (s >> 20) & 1u; }
__host__ __device__ inline u32 mk_state(u32 turn, u32 castle, u32 ep, u32 clock, u32 epl)
{
    return turn | (castle << 1) | (ep << 5) | (clock << 12) | (epl << 20);
}

constexpr u64 FILE_A = 0x0101010101010101ull;
constexpr u64 FILE_B = FILE_A << 1;
constexpr u64 FILE_G = FILE_A << 6;
constexpr u64 FILE_H = FILE_A << 7;
constexpr u64 RANK_1 = 0xFFull;
constexpr u64 RANK_8 = 0xFFull << 56;
constexpr u64 DIAG_MAIN = 0x8040201008040201ull;  // a1-h8
constexpr u64 DIAG_ANTI = 0x0102040810204080ull;  // h1-a8
constexpr u64 DARK_SQ = 0xAA55AA55AA55AA55ull;
constexpr u64 LIGHT_SQ = 0x55AA55AA55AA55AAull;

__device__ inline u64 bit(int sq) { return 1ull << sq; }
__device__ inline int msb(u64 x) { return 63 - __builtin_clzll(x); }
__device__ inline int lsb(u64 x) { return __builtin_ctzll(x); }
__device__ inline int popc(u64 x) { return __builtin_popcountll(x); }

__device__ inline u64 occupied(const Board &b)
{
    return b.bb[0] | b.bb[1] | b.bb[2] | b.bb[3] | b.bb[4] | b.bb[5];
}

// ---- line masks through a square ---------------------------------------------------
__device__ inline u64 rank_mask(int sq) { return RANK_1 << (sq & 56); }
__device__ inline u64 file_mask(int sq) { return FILE_A << (sq & 7); }
__device__ inline u64 diag_mask(int sq)
{
    int d = (sq >> 3) - (sq & 7);
    return d >= 0 ? DIAG_MAIN << (8 * d) : DIAG_MAIN >> (8 * -d);
}
__device__ inline u64 anti_mask(int sq)
{
    int s = (sq >> 3) + (sq & 7) - 7;
    return s >= 0 ? DIAG_ANTI << (8 * s) : DIAG_ANTI >> (8 * -s);
}

// hyperbola quintessence along one line mask; `occ` may or may not contain sq
__device__ inline u64 line_attacks(int sq, u64 occ, u64 mask)
{
    u64 r = bit(sq);
    u64 o = (occ & mask) | r;     // o - 2r needs the slider's own bit in o
    u64 fwd = o - 2 * r;
    u64 rev = __builtin_bitreverse64(__builtin_bitreverse64(o) - 2 * __builtin_bitreverse64(r));
    return (fwd ^ rev) & mask;
}
__device__ inline u64 rook_attacks(int sq, u64 occ)
{
    return line_attacks(sq, occ, rank_mask(sq)) | line_attacks(sq, occ, file_mask(sq));
}
__device__ inline u64 bishop_attacks(int sq, u64 occ)
{
    return line_attacks(sq, occ, diag_mask(sq)) | line_attacks(sq, occ, anti_mask(sq));
}
__device__ inline u64 knight_attacks(int sq)
{
    u64 b = bit(sq);
    return ((b << 17) & ~FILE_A) | ((b << 15) & ~FILE_H) | ((b << 10) & ~(FILE_A | FILE_B)) |
           ((b << 6) & ~(FILE_G | FILE_H)) | ((b >> 17) & ~FILE_H) | ((b >> 15) & ~FILE_A) |
           ((b >> 10) & ~(FILE_G | FILE_H)) | ((b >> 6) & ~(FILE_A | FILE_B));
}
__device__ inline u64 king_attacks(int sq)
{
    u64 b = bit(sq);
    u64 h = ((b << 1) & ~FILE_A) | ((b >> 1) & ~FILE_H);
    u64 row = h | b;
    return h | (row << 8) | (row >> 8);
}
// squares attacked by a pawn of colour `white` standing on sq
__device__ inline u64 pawn_attacks(int sq, bool white)
{
    u64 b = bit(sq);
    return white ? (((b << 9) & ~FILE_A) | ((b << 7) & ~FILE_H))
                 : (((b >> 7) & ~FILE_A) | ((b >> 9) & ~FILE_H));
}

// pieces of side `by_white` attacking `sq` under occupancy `occ`
__device__ inline u64 attackers_to(const Board &b, int sq, u64 occ, bool by_white)
{
    u64 side = by_white ? b.white : ~b.white;
    u64 rq = b.bb[ROOK] | b.bb[QUEEN], bq = b.bb[BISHOP] | b.bb[QUEEN];
    u64 a = (rook_attacks(sq, occ) & rq) | (bishop_attacks(sq, occ) & bq) |
            (knight_attacks(sq) & b.bb[KNIGHT]) | (king_attacks(sq) & b.bb[KING]) |
            (pawn_attacks(sq, !by_white) & b.bb[PAWN]);
    return a & side & occ;
}

// piece type on sq (0..5) or -1
__device__ inline int piece_at(const Board &b, int sq)
{
    u64 m = bit(sq);
    int t = -1;
#pragma unroll
    for (int i = 0; i < 6; i++) t = (b.bb[i] & m) ? i : t;
    return t;
}

// ---- python-chess Board.push for a LEGAL move (uniform) ----------------------------------
// The derived "legal en-passant exists" bit of the result is left 0; the caller sets it
// after running move generation on the new position.
__device__ inline Board apply_move(const Board &b, u32 mv)
{
    Board n = b;
    int from = mv & 63, to = (mv >> 6) & 63, promo = (mv >> 12) & 7;
    u64 fb = bit(from), tb = bit(to);
    bool white = st_turn(b.state);
    int pt = piece_at(b, from);
    u64 occ = occupied(b);
    bool capture = (occ & tb) != 0;
    u32 castle = st_castle(b.state), ep = st_ep(b.state), clock = st_clock(b.state);
    u32 new_ep = NO_EP;
    clock = (pt == PAWN || capture) ? 0u : (clock < 255u ? clock + 1u : 255u);
    if ((fb | tb) & bit(7)) castle &= ~CR_WK;
    if ((fb | tb) & bit(0)) castle &= ~CR_WQ;
    if ((fb | tb) & bit(63)) castle &= ~CR_BK;
    if ((fb | tb) & bit(56)) castle &= ~CR_BQ;
    if (pt == KING) castle &= white ? ~(CR_WK | CR_WQ) : ~(CR_BK | CR_BQ);
    // remove any captured piece, lift the mover
#pragma unroll
    for (int i = 0; i < 6; i++) n.bb[i] &= ~(tb | fb);
    n.white &= ~(tb | fb);
    if (pt == PAWN) {
        int diff = to - from;
        if (diff == 16 && (from >> 3) == 1) new_ep = from + 8;
        else if (diff == -16 && (from >> 3) == 6) new_ep = from - 8;
        else if ((u32)to == ep && !capture && (diff == 7 || diff == 9 || diff == -7 || diff == -9)) {
            u64 cb = bit(white ? to - 8 : to + 8);
            n.bb[PAWN] &= ~cb;
            n.white &= ~cb;
        }
    }
    if (pt == KING && (to - from == 2 || from - to == 2)) {
        int rf = to > from ? from + 3 : from - 4, rt = to > from ? from + 1 : from - 1;
        n.bb[ROOK] = (n.bb[ROOK] & ~bit(rf)) | bit(rt);
        if (white) n.white = (n.white & ~bit(rf)) | bit(rt);
    }
    int np = promo ? promo - 1 : pt;      // promo codes 2..5 = N,B,R,Q -> index 1..4
#pragma unroll
    for (int i = 0; i < 6; i++) n.bb[i] |= (i == np) ? tb : 0ull;
    if (white) n.white |= tb;
    n.state = mk_state(white ? 0u : 1u, castle, new_ep, clock, 0u);
    n.pad = 0;
    return n;
}

// ---- python-chess is_insufficient_material (both sides has_insufficient_material) -----------
__device__ inline bool side_insufficient(const Board &b, bool white)
{
    u64 occ = occupied(b);
    u64 own = white ? b.white : (occ & ~b.white), opp = occ & ~own;
    if (own & (b.bb[PAWN] | b.bb[ROOK] | b.bb[QUEEN])) return false;
    if (own & b.bb[KNIGHT]) return popc(own) <= 2 && !(opp & ~b.bb[KING] & ~b.bb[QUEEN]);
    if (own & b.bb[BISHOP]) {
        bool same = !(b.bb[BISHOP] & DARK_SQ) || !(b.bb[BISHOP] & LIGHT_SQ);
        return same && !b.bb[PAWN] && !b.bb[KNIGHT];
    }
    return true;
}
__device__ inline bool insufficient_material(const Board &b)
{
    return side_insufficient(b, true) && side_insufficient(b, false);
}

// ---- python-chess _transposition_key as a 64-bit filter hash + exact comparison ------------
__device__ inline u64 mix64(u64 h, u64 w)
{
    h ^= w;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return h;
}
__device__ inline u32 key_bits(u32 state)
{
    u32 ep = st_eplegal(state) ? st_ep(state) : (u32)NO_EP;
    return st_turn(state) | (st_castle(state) << 1) | (ep << 5);
}
__device__ inline u64 board_hash(const Board &b)
{
    u64 h = 0x243F6A8885A308D3ull;
#pragma unroll
    for (int i = 0; i < 6; i++) h = mix64(h, b.bb[i]);
    h = mix64(h, b.white);
    h = mix64(h, key_bits(b.state));
    return h;
}
__device__ inline bool same_key(const Board &a, const Board &b)
{
    bool eq = a.white == b.white && key_bits(a.state) == key_bits(b.state);
#pragma unroll
    for (int i = 0; i < 6; i++) eq = eq && a.bb[i] == b.bb[i];
    return eq;
}

// Game.get_result (game.py:92-109) from the facts about a position.
// n_legal: number of legal moves; rep: occurrences of the position incl. itself.
__device__ inline int position_result(const Board &b, int n_legal, bool in_check, int rep)
{
    u32 clock = st_clock(b.state);
    if (clock >= 100 && n_legal > 0) return 0;                 // can_claim_fifty_moves
    bool over = n_legal == 0 || insufficient_material(b) || rep >= 5 ||
                (clock >= 150 && n_legal > 0);
    if (!over) return RESULT_NONE;
    if (n_legal == 0 && in_check) return st_turn(b.state) ? -1 : 1;
    return 0;
}

}  // namespace crl
