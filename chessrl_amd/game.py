"""``Game`` -- host mirror of the reference's rules wrapper, backed by the HIP library.

Same surface as /root/reference/src/chessrl/game.py:11-112 (``move``,
``get_legal_moves``, ``get_result``, ``get_copy``, ``get_history``, ``get_fen``,
``set_fen``, ``turn``, ``reset``, ``free``, ``len()``; ``NULL_MOVE``, ``WHITE``,
``BLACK``), but every rule decision is taken by the gfx950 kernels through the
C-ABI (crl_legal_moves / crl_push_moves / crl_results): there is no
python-chess and no CPU rules fallback.  A ``Game`` owns one slot of a shared
device arena; ``get_copy`` is a device-side deep copy (board and move stack).

This object-per-game surface exists for drop-in compatibility; throughput
comes from the batched path (``selfplay.SelfPlayRunner`` / ``LockstepEngine``).
``plot_board`` (game.py:114-135, debug rendering) is out of scope.
"""
from datetime import datetime

import numpy as np

from . import _lib

ARENA_SLOTS = 4096
ARENA_MAX_PLIES = 4096
_FILES = "abcdefgh"
_PROMO = " pnbrqk"
NO_EP = 64


def move_to_uci(m):
    m = int(m)
    f, t, p = m & 63, (m >> 6) & 63, (m >> 12) & 7
    s = _FILES[f & 7] + str((f >> 3) + 1) + _FILES[t & 7] + str((t >> 3) + 1)
    return s + (_PROMO[p] if p else "")


def uci_to_move(u):
    """UCI string -> move id, or None when it cannot be a move ('00000', garbage)."""
    if not isinstance(u, str) or len(u) not in (4, 5):
        return None
    f0, r0, f1, r1 = _FILES.find(u[0]), "12345678".find(u[1]), _FILES.find(u[2]), "12345678".find(u[3])
    if min(f0, r0, f1, r1) < 0:
        return None
    p = 0
    if len(u) == 5:
        p = "nbrq".find(u[4]) + 2
        if p < 2:
            return None
    return (r0 * 8 + f0) | ((r1 * 8 + f1) << 6) | (p << 12)


def clean_castling_rights(cas, kings, rooks, white):
    """python-chess ``Board.clean_castling_rights()`` (standard chess), which ``chess.Board(fen)`` applies wherever
    it uses the rights -- move generation, ``push`` and the transposition key of the repetition rules
    (game.py:17-21,92-109): a right survives only with its king on e1 / e8 and a rook of that colour on its
    corner.  The device hashes the bits as they stand (csrc/board.hpp: key_bits), so a FEN claiming more than the
    position holds is cleaned HERE: otherwise a king's first move would change the key where python-chess sees
    the same position again, and fivefold counts would differ.  cas: K=1, Q=2, k=4, q=8."""
    black = ~white
    wk, bk = (kings & white) >> 4 & 1, (kings & black) >> 60 & 1
    wr, br = rooks & white, rooks & black
    keep = 0
    keep |= 1 if (wk and (wr >> 7) & 1) else 0
    keep |= 2 if (wk and wr & 1) else 0
    keep |= 4 if (bk and (br >> 63) & 1) else 0
    keep |= 8 if (bk and (br >> 56) & 1) else 0
    return cas & keep


def board_row_from_fen(fen):
    """FEN (piece placement, optionally the full record) -> np.uint64[8] crl_board row."""
    parts = fen.split()
    row = np.zeros(8, dtype=np.uint64)
    bb = [0] * 6
    white = 0
    for r, line in enumerate(parts[0].split("/")):
        f = 0
        for ch in line:
            if ch.isdigit():
                f += int(ch)
                continue
            sq = (7 - r) * 8 + f
            bb["pnbrqk".index(ch.lower())] |= 1 << sq
            if ch.isupper():
                white |= 1 << sq
            f += 1
    turn = 0 if (len(parts) > 1 and parts[1] == "b") else 1
    cas = 0
    if len(parts) > 2:
        cas = sum(bit for ch, bit in (("K", 1), ("Q", 2), ("k", 4), ("q", 8)) if ch in parts[2])
        cas = clean_castling_rights(cas, bb[5], bb[3], white)
    ep = NO_EP
    if len(parts) > 3 and parts[3] != "-":
        ep = _FILES.index(parts[3][0]) + 8 * (int(parts[3][1]) - 1)
    clock = min(int(parts[4]), 255) if len(parts) > 4 else 0
    for i in range(6):
        row[i] = bb[i]
    row[6] = white
    row[7] = turn | (cas << 1) | (ep << 5) | (clock << 12)
    return row


def board_fen_from_row(row):
    """python-chess ``board_fen()``: piece placement only (game.py:68-69)."""
    out = []
    for rank in range(7, -1, -1):
        s, empty = "", 0
        for f in range(8):
            sq = rank * 8 + f
            ch = None
            for t in range(6):
                if (int(row[t]) >> sq) & 1:
                    ch = "pnbrqk"[t]
            if ch is None:
                empty += 1
                continue
            if empty:
                s += str(empty)
                empty = 0
            s += ch.upper() if (int(row[6]) >> sq) & 1 else ch
        out.append(s + (str(empty) if empty else ""))
    return "/".join(out)


class _Arena(object):
    """A shared crl_ctx whose slots are handed to Game objects."""

    def __init__(self, device=0):
        self.ctx = _lib.Context(ARENA_SLOTS, 1, max_plies=ARENA_MAX_PLIES, device=device)
        self.free = list(range(ARENA_SLOTS - 1, -1, -1))
        self.device = device
        self._planes = None

    def alloc(self):
        if not self.free:
            raise RuntimeError("Game arena exhausted (%d live Game objects); free() some" % ARENA_SLOTS)
        return self.free.pop()

    def release(self, slot):
        self.free.append(slot)

    def one(self, slot):
        self.ctx.set_window(slot, 1)
        return self.ctx

    def planes(self):
        import torch
        if self._planes is None:
            self._planes = torch.zeros((1, 8, 8, _lib.PLANES), dtype=torch.float16,
                                       device=torch.device("cuda", self.device))
        return self._planes


_arena = None


def arena():
    global _arena
    if _arena is None:
        _arena = _Arena()
    return _arena


class Game(object):

    NULL_MOVE = "00000"
    WHITE = True
    BLACK = False

    def __init__(self, board=None, player_color=True, date=None):
        """``board``: None (standard start), a FEN string, a ``crl_board`` row, another ``Game`` to
        deep-copy, or -- what the reference passes (game.py:17-21) -- a python-chess ``chess.Board``,
        recognised by duck typing (``root()``, ``fen()``, ``move_stack`` of moves with ``uci()``; nothing
        is imported): the game starts from the board's root position and its move stack is replayed
        through the rules kernels, so history planes and repetition counts are the board's."""
        a = arena()
        self._slot = a.alloc()
        if isinstance(board, Game):
            a.ctx.copy_game(self._slot, board._slot)
        elif isinstance(board, str):
            a.one(self._slot).set_positions(board_row_from_fen(board)[None])
        elif board is None:
            a.one(self._slot).reset_games()
        elif callable(getattr(board, "fen", None)) and hasattr(board, "move_stack"):
            root = board.root() if callable(getattr(board, "root", None)) else None
            stack = list(board.move_stack)
            if root is None and stack:
                a.release(self._slot)
                self._slot = None
                raise TypeError("a board with a move stack must offer root() (python-chess does)")
            a.one(self._slot).set_positions(board_row_from_fen((root or board).fen())[None])
            for mv in stack:
                if not self.move(mv.uci() if callable(getattr(mv, "uci", None)) else str(mv)):
                    a.release(self._slot)
                    self._slot = None
                    raise ValueError("move stack of the board does not replay: %s" % mv)
        else:
            a.one(self._slot).set_positions(np.asarray(board, dtype=np.uint64).reshape(1, 8))
        self.player_color = player_color
        self.date = date
        if self.date is None:
            self.date = datetime.now().strftime("%d/%m/%Y %H:%M:%S")

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def _ctx(self):
        if self._slot is None:
            raise RuntimeError("Game was freed")
        return arena().one(self._slot)

    # ---- reference API ---------------------------------------------------------------------
    def move(self, movement):
        """Apply a UCI move iff it is legal; returns success (game.py:28-41)."""
        m = uci_to_move(movement)
        if m is None:
            return False
        return bool(self._ctx().push_moves(np.array([m], dtype=np.uint16))[0])

    def legal_move_ids(self):
        moves, counts = self._ctx().legal_moves()
        return moves[0, :counts[0]].copy()

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

    def move_ids(self):
        moves, plies, _ = self._ctx().records()
        return moves[0, :plies[0]].copy()

    def get_history(self):
        return {"moves": [move_to_uci(m) for m in self.move_ids()],
                "result": self.get_result(),
                "player_color": self.player_color,
                "date": self.date}

    def board_row(self):
        return self._ctx().get_positions(1)[0]

    def get_fen(self):
        return board_fen_from_row(self.board_row())

    def set_fen(self, fen):
        """python-chess ``set_board_fen``: replaces the piece placement only.  Deviation: the
        device slot restarts its move stack (the reference keeps a now-inconsistent one)."""
        row = self.board_row()
        new = board_row_from_fen(fen)
        row[:7] = new[:7]
        self._ctx().set_positions(row[None])

    @property
    def turn(self):
        return bool(int(self.board_row()[7]) & 1)

    def get_copy(self):
        return Game(board=self)

    def reset(self):
        self._ctx().reset_games()

    def free(self):
        if getattr(self, "_slot", None) is not None and _arena is not None:
            _arena.release(self._slot)
        self._slot = None

    def get_result(self):
        """1 / -1 / 0 for white, None while the game runs (game.py:92-109)."""
        r = int(self._ctx().results()[0])
        return None if r == _lib.RESULT_NONE else r

    def __len__(self):
        return int(self._ctx().records(with_moves=False)[1][0])

    def plot_board(self, save_path=None):
        raise NotImplementedError("debug rendering (game.py:114-135) is out of scope of this path")
