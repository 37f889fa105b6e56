"""oracle/encoder_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's board encoder and move-label table
(/root/reference/src/chessrl/netencoder.py:13-134), operating on the C
oracle's positions instead of python-chess boards.

PARITY STATUS: ``get_uci_labels`` is pinned by a golden fixture produced from
the reference's own function (tests/golden/uci_labels.json, made by
oracle/make_golden.py).  ``get_game_state`` needs python-chess
(netencoder.py:25-27,58-67), which is absent here: it is pinned by
tests/golden/encoder_cases.json = the reference's own get_game_state and helpers
executed over an adapter of the four python-chess members they touch
(oracle/ref_loader.py: pieces, mirror, tolist, pop) -- the layout is the
reference's, the reading of those four primitives is ours.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np


def get_uci_labels():
    """1968 UCI labels; restates netencoder.py:94-134 (order is the contract)."""
    labels = []
    letters = "abcdefgh"
    numbers = "12345678"
    for l1 in range(8):
        for n1 in range(8):
            dests = ([(t, n1) for t in range(8)] + [(l1, t) for t in range(8)] +
                     [(l1 + t, n1 + t) for t in range(-7, 8)] +
                     [(l1 + t, n1 - t) for t in range(-7, 8)] +
                     [(l1 + a, n1 + b) for (a, b) in
                      [(-2, -1), (-1, -2), (-2, 1), (1, -2), (2, -1), (-1, 2), (2, 1), (1, 2)]])
            for (l2, n2) in dests:
                if (l1, n1) != (l2, n2) and 0 <= l2 < 8 and 0 <= n2 < 8:
                    labels.append(letters[l1] + numbers[n1] + letters[l2] + numbers[n2])
    for l1 in range(8):
        le = letters[l1]
        for p in "qrbn":
            labels.append(le + "2" + le + "1" + p)
            labels.append(le + "7" + le + "8" + p)
            if l1 > 0:
                ll = letters[l1 - 1]
                labels.append(le + "2" + ll + "1" + p)
                labels.append(le + "7" + ll + "8" + p)
            if l1 < 7:
                lr = letters[l1 + 1]
                labels.append(le + "2" + lr + "1" + p)
                labels.append(le + "7" + lr + "8" + p)
    return labels


def _bb_to_plane(bb):
    """bitboard -> (8,8) array, row 0 = rank 8, col 0 = file a.

    netencoder.py:25-26: ``board.pieces(i, color).mirror().tolist()`` reshaped
    (8,8): mirror() flips ranks, tolist() walks squares a1..h8.
    """
    bits = np.array([(int(bb) >> sq) & 1 for sq in range(64)], dtype=np.float64).reshape(8, 8)
    return bits[::-1, :]


def _pieces_one_hot(b, color):
    """netencoder.py:13-30: 7 planes {blank, P, N, B, R, Q, K} for one colour."""
    occ = 0
    for t in range(6):
        occ |= int(b.bb[t])
    own = int(b.white) if color else (occ & ~int(b.white))
    mask = np.zeros((8, 8, 7))
    for t in range(6):
        mask[:, :, t + 1] = _bb_to_plane(int(b.bb[t]) & own)
    mask[:, :, 0] = (~np.array(mask.sum(axis=-1), dtype=bool)).astype(int)
    return mask


def _current_state(b):
    """netencoder.py:33-44: black planes then white planes (14 channels)."""
    return np.concatenate((_pieces_one_hot(b, False), _pieces_one_hot(b, True)), axis=-1)


def get_game_state(game, flipped=False, T=8):
    """netencoder.py:72-91 on an OracleGame: (8,8,127) float64."""
    current = _current_state(game.board_at(0))
    history = np.zeros((8, 8, 14 * T))
    n = len(game)
    for i in range(T):                      # netencoder.py:59-67 (pop until IndexError)
        if i + 1 > n:
            break
        history[:, :, i * 14:(i + 1) * 14] = _current_state(game.board_at(i + 1))
    turn = np.full((8, 8, 1), game.turn, dtype=float)
    out = np.concatenate((current, history, turn), axis=-1)
    if flipped:
        out = np.rot90(out, k=2)
    return out
