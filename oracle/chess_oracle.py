"""oracle/chess_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes wrapper over oracle/chess_oracle.c that duck-types the reference's
``Game`` (/root/reference/src/chessrl/game.py:11-112) closely enough for the
reference's own ``mctree.py`` to run on it (it touches ``get_legal_moves``,
``move``, ``get_result``, ``get_copy``, ``turn`` and ``board.move_stack``,
mctree.py:31,186-188,241-246,308).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Parity status of the rules: see the header of chess_oracle.c
("parity unpinned" against python-chess 0.28.3 for move ORDER).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "chess_oracle.c")
_SO = os.path.join(_HERE, "_build", "libchess_oracle.so")

NO_EP = 64
RESULT_NONE = 2
NULL_MOVE = "00000"


class OcBoard(ctypes.Structure):
    _fields_ = [("bb", ctypes.c_uint64 * 6), ("white", ctypes.c_uint64),
                ("state", ctypes.c_uint32), ("pad", ctypes.c_uint32)]


def build(force=False):
    """Compile the C restatement with gcc (the checker, never the product)."""
    if (not force and os.path.exists(_SO) and os.path.exists(_SRC)
            and os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    if not os.path.exists(_SRC):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, _SRC])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.og_new.restype = ctypes.c_void_p
        L.og_from_board.restype = ctypes.c_void_p
        L.og_from_board.argtypes = [ctypes.POINTER(OcBoard)]
        L.og_copy.restype = ctypes.c_void_p
        L.og_copy.argtypes = [ctypes.c_void_p]
        L.og_free.argtypes = [ctypes.c_void_p]
        for n in ("og_ply", "og_turn", "og_result", "og_repetitions", "og_in_check",
                  "og_insufficient"):
            getattr(L, n).argtypes = [ctypes.c_void_p]
            getattr(L, n).restype = ctypes.c_int
        L.og_board.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(OcBoard)]
        L.og_move_at.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.og_move_at.restype = ctypes.c_uint16
        L.og_legal_moves.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint16)]
        L.og_legal_moves.restype = ctypes.c_int
        L.og_push.argtypes = [ctypes.c_void_p, ctypes.c_uint16]
        L.og_push.restype = ctypes.c_int
        L.og_perft.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.og_perft.restype = ctypes.c_uint64
        _lib = L
    return _lib


# ---- move <-> UCI ------------------------------------------------------------
_PROMO = " pnbrqk"


def move_to_uci(m):
    m = int(m)
    f, t, p = m & 63, (m >> 6) & 63, (m >> 12) & 7
    s = "abcdefgh"[f & 7] + str((f >> 3) + 1) + "abcdefgh"[t & 7] + str((t >> 3) + 1)
    return s + (_PROMO[p] if p else "")


def uci_to_move(u):
    """UCI string -> u16 move id, or None for anything malformed ('00000')."""
    if not isinstance(u, str) or len(u) not in (4, 5):
        return None
    try:
        f = "abcdefgh".index(u[0]) + 8 * (int(u[1]) - 1)
        t = "abcdefgh".index(u[2]) + 8 * (int(u[3]) - 1)
        p = _PROMO.index(u[4]) if len(u) == 5 else 0
    except ValueError:
        return None
    if not (0 <= f < 64 and 0 <= t < 64) or (len(u) == 5 and p < 2):
        return None
    return f | (t << 6) | (p << 12)


# ---- FEN helpers (tests only) -----------------------------------------------
def board_from_fen(fen):
    parts = fen.split()
    b = OcBoard()
    rows = parts[0].split("/")
    for r, row in enumerate(rows):
        rank = 7 - r
        f = 0
        for ch in row:
            if ch.isdigit():
                f += int(ch)
                continue
            sq = rank * 8 + f
            b.bb["pnbrqk".index(ch.lower())] |= 1 << sq
            if ch.isupper():
                b.white |= 1 << sq
            f += 1
    turn = 1 if len(parts) < 2 or parts[1] == "w" else 0
    cas = 0
    if len(parts) > 2:
        for ch, bit in (("K", 1), ("Q", 2), ("k", 4), ("q", 8)):
            if ch in parts[2]:
                cas |= bit
    # chess.Board(fen) uses castling rights only through clean_castling_rights() (python-chess 0.28.3, standard
    # chess: a right needs its king on e1 / e8 and an own rook on its corner) -- in move generation, in push and
    # in the transposition key behind is_fivefold_repetition (reference call sites: game.py:17-21,92-109)
    kings, rooks, black = b.bb[5], b.bb[3], ~b.white
    if not ((kings & b.white) >> 4 & 1 and (rooks & b.white) >> 7 & 1):
        cas &= ~1
    if not ((kings & b.white) >> 4 & 1 and (rooks & b.white) & 1):
        cas &= ~2
    if not ((kings & black) >> 60 & 1 and (rooks & black) >> 63 & 1):
        cas &= ~4
    if not ((kings & black) >> 60 & 1 and (rooks & black) >> 56 & 1):
        cas &= ~8
    ep = NO_EP
    if len(parts) > 3 and parts[3] != "-":
        ep = "abcdefgh".index(parts[3][0]) + 8 * (int(parts[3][1]) - 1)
    clock = int(parts[4]) if len(parts) > 4 else 0
    b.state = turn | (cas << 1) | (ep << 5) | (min(clock, 255) << 12)
    return b


def board_fen(b):
    """Piece-placement FEN (python-chess board_fen(), game.py:68-69)."""
    rows = []
    for rank in range(7, -1, -1):
        row, empty = "", 0
        for f in range(8):
            sq = rank * 8 + f
            ch = None
            for t in range(6):
                if (b.bb[t] >> sq) & 1:
                    ch = "pnbrqk"[t]
            if ch is None:
                empty += 1
                continue
            if empty:
                row += str(empty)
                empty = 0
            row += ch.upper() if (b.white >> sq) & 1 else ch
        if empty:
            row += str(empty)
        rows.append(row)
    return "/".join(rows)


class _Move(object):
    """Stand-in for chess.Move: str() is the UCI string (mctree.py:196)."""
    __slots__ = ("m",)

    def __init__(self, m):
        self.m = int(m)

    def uci(self):
        return move_to_uci(self.m)

    __str__ = uci

    def __repr__(self):
        return "Move(%s)" % self.uci()


class _MoveStack(object):
    """Lazy view of the C move list supporting len() and (negative) indexing."""

    def __init__(self, h):
        self._h = h

    def __len__(self):
        return lib().og_ply(self._h)

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("move stack index out of range")
        return _Move(lib().og_move_at(self._h, i))

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _BoardView(object):
    def __init__(self, game):
        self._g = game

    @property
    def move_stack(self):
        return _MoveStack(self._g._h)

    @property
    def turn(self):
        return bool(lib().og_turn(self._g._h))


class OracleGame(object):
    """Duck-typed reference ``Game`` backed by the C oracle."""

    NULL_MOVE = NULL_MOVE
    WHITE = True
    BLACK = False

    def __init__(self, board=None, player_color=True, date=None, _handle=None):
        if _handle is not None:
            self._h = _handle
        elif board is not None:
            self._h = lib().og_from_board(ctypes.byref(board))
        else:
            self._h = lib().og_new()
        self.player_color = player_color
        self.date = date
        self.board = _BoardView(self)

    def __del__(self):
        try:
            lib().og_free(self._h)
        except Exception:
            pass

    # -- reference Game API (game.py) ------------------------------------------
    def move(self, movement):
        m = uci_to_move(movement)
        if m is None:
            return False
        return bool(lib().og_push(self._h, m))

    def legal_move_ids(self):
        buf = (ctypes.c_uint16 * 256)()
        n = lib().og_legal_moves(self._h, buf)
        return [buf[i] for i in range(n)]

    def get_legal_moves(self, final_states=False):
        moves = [move_to_uci(m) for m in self.legal_move_ids()]
        if final_states:
            states = []
            for m in moves:
                gi = self.get_copy()
                gi.move(m)
                states.append(gi)
            moves = (moves, states)
        return moves

    def get_result(self):
        r = lib().og_result(self._h)
        return None if r == RESULT_NONE else r

    def get_copy(self):
        return OracleGame(_handle=lib().og_copy(self._h))

    def get_history(self):
        return {"moves": [m.uci() for m in self.board.move_stack],
                "result": self.get_result(),
                "player_color": self.player_color,
                "date": self.date}

    def get_fen(self):
        return board_fen(self.board_at(0))

    @property
    def turn(self):
        return bool(lib().og_turn(self._h))

    def __len__(self):
        return lib().og_ply(self._h)

    # -- oracle extras -----------------------------------------------------------
    def board_at(self, back=0):
        b = OcBoard()
        lib().og_board(self._h, back, ctypes.byref(b))
        return b

    def perft(self, depth):
        return int(lib().og_perft(self._h, depth))

    def repetitions(self):
        return lib().og_repetitions(self._h)


def board_to_array(b):
    """OcBoard -> np.uint64[8] (bb[0..5], white, state|pad) for the C-ABI."""
    a = np.zeros(8, dtype=np.uint64)
    for i in range(6):
        a[i] = b.bb[i]
    a[6] = b.white
    a[7] = int(b.state)
    return a
