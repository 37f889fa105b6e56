"""CPU: the C chess oracle against public known answers (perft suite, python-chess README
listing) and hand-derived rule cases.  The oracle's move ORDER is otherwise unpinned against
python-chess 0.28.3 (absent from this image) -- see oracle/chess_oracle.c."""
import pytest

from oracle.chess_oracle import (OracleGame, board_fen, board_from_fen, move_to_uci,
                                 uci_to_move)
from tests.util import PERFT_FENS

# https://www.chessprogramming.org/Perft_Results (public known answers)
PERFT = {
    "start": ("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", [20, 400, 8902, 197281]),
    "kiwipete": (PERFT_FENS["kiwipete"], [48, 2039, 97862]),
    "pos3": (PERFT_FENS["pos3"], [14, 191, 2812, 43238, 674624]),
    "pos4": (PERFT_FENS["pos4"], [6, 264, 9467, 422333]),
    "pos4m": (PERFT_FENS["pos4m"], [6, 264, 9467, 422333]),
    "pos5": (PERFT_FENS["pos5"], [44, 1486, 62379]),
    "pos6": (PERFT_FENS["pos6"], [46, 2079, 89890]),
}


@pytest.mark.parametrize("name", sorted(PERFT))
def test_perft_known_answers(name):
    fen, counts = PERFT[name]
    g = OracleGame(board=board_from_fen(fen))
    assert [g.perft(d + 1) for d in range(len(counts))] == counts


def test_start_position_order_matches_python_chess_listing():
    # python-chess README: <LegalMoveGenerator ... (Nh3, Nf3, Nc3, Na3, h3, g3, ..., a4)>
    assert OracleGame().get_legal_moves() == [
        "g1h3", "g1f3", "b1c3", "b1a3", "h2h3", "g2g3", "f2f3", "e2e3", "d2d3", "c2c3", "b2b3",
        "a2a3", "h2h4", "g2g4", "f2f4", "e2e4", "d2d4", "c2c4", "b2b4", "a2a4"]


def test_generation_order_categories():
    """pieces (from high->low, to high->low), castling K then Q, pawn captures with promotions
    Q,R,B,N, single pushes, double pushes, en passant last."""
    g = OracleGame(board=board_from_fen("r3k2r/1P6/8/3pP3/8/8/P7/R3K2R w KQkq d6 0 2"))
    mv = g.get_legal_moves()
    assert mv.index("e1g1") + 1 == mv.index("e1c1")                     # king side first
    assert mv.index("e1c1") < mv.index("b7a8q")                         # castling before pawns
    i = mv.index("b7a8q")
    assert mv[i:i + 4] == ["b7a8q", "b7a8r", "b7a8b", "b7a8n"]          # promotion order
    assert mv.index("b7a8n") < mv.index("b7b8q") < mv.index("e5e6") < mv.index("a2a3")
    assert mv.index("a2a3") < mv.index("a2a4") < mv.index("e5d6")       # double pushes, then ep
    assert mv[-1] == "e5d6"
    assert mv[0] == "h1h8" and mv[1] == "h1h7"                          # highest from-square first


def test_evasions_king_first_then_blocks():
    g = OracleGame(board=board_from_fen("4k3/8/8/8/7b/8/3N4/R3K3 w Q - 0 1"))
    mv = g.get_legal_moves()
    assert mv[:3] == ["e1e2", "e1f1", "e1d1"] and mv[3:] == ["d2f2"] or mv == ["e1e2", "e1f1", "e1d1"]
    assert "e1c1" not in mv                                             # no castling out of check


def test_en_passant_legality():
    g = OracleGame(board=board_from_fen(PERFT_FENS["ep_pin"]))           # capture would expose the king
    assert "b5c6" not in g.get_legal_moves()
    assert (g.board_at(0).state >> 20) & 1 == 0
    g = OracleGame(board=board_from_fen(PERFT_FENS["ep_check"]))         # ep captures the checking pawn
    assert "e4d3" in g.get_legal_moves() and (g.board_at(0).state >> 20) & 1 == 1


def test_push_semantics():
    g = OracleGame()
    assert not g.move("e2e5") and not g.move("00000") and not g.move("e7e5") and len(g) == 0
    for u in ["e2e4", "e7e5", "g1f3", "b8c6", "f1c4", "g8f6", "e1g1"]:
        assert g.move(u)
    b = g.board_at(0)
    assert board_fen(b) == "r1bqkb1r/pppp1ppp/2n2n2/4p3/2B1P3/5N2/PPPP1PPP/RNBQ1RK1"
    assert (b.state >> 1) & 15 == 0b1100                                # white rights gone
    assert (b.state >> 12) & 255 == 5 and b.state & 1 == 0             # clock, black to move
    assert uci_to_move("e7e8q") == 52 | (60 << 6) | (5 << 12) and move_to_uci(52 | (60 << 6) | (2 << 12)) == "e7e8n"
    assert [m.uci() for m in g.board.move_stack][-2:] == ["g8f6", "e1g1"]
    c = g.get_copy()
    assert c.move("f8c5") and len(c) == 8 and len(g) == 7               # deep copy


def test_results():
    def res(fen):
        return OracleGame(board=board_from_fen(fen)).get_result()
    assert res("7k/6Q1/6K1/8/8/8/8/8 b - - 0 1") == 1                    # white mates
    assert res("8/8/8/8/8/6k1/6q1/7K w - - 0 1") == -1                   # black mates
    assert res("7k/5Q2/6K1/8/8/8/8/8 b - - 0 1") == 0                    # stalemate
    assert res("8/8/8/4k3/8/8/4K3/7B w - - 0 1") == 0                    # K+B v K
    assert res("8/8/8/4k3/8/8/4K3/6NN w - - 0 1") is None                # K+N+N v K
    assert res("8/8/8/4k3/8/8/4K3/7R w - - 99 1") is None
    assert res("8/8/8/4k3/8/8/4K3/7R w - - 100 1") == 0                  # fifty-move claim
    assert res("7k/6Q1/6K1/8/8/8/8/8 b - - 100 1") == 1                  # mate beats the clock
    g = OracleGame()
    for _ in range(4):
        assert g.get_result() is None
        for u in ["g1f3", "g8f6", "f3g1", "f6g8"]:
            g.move(u)
    assert g.repetitions() == 5 and g.get_result() == 0                 # fivefold, not threefold


# FENs that claim castling rights the position does not hold -> what chess.Board(fen).clean_castling_rights() keeps
# (python-chess 0.28.3, standard chess: king on e1 / e8 AND an own rook on the corner)
UNCLEAN_FENS = [
    ("4k3/p7/8/8/8/8/P7/4K3 w KQkq - 0 1", ""),              # no rooks at all
    ("4k3/8/8/8/8/8/8/4K2R b KQkq - 0 1", "K"),              # VERDICT r4's example: only h1 holds a rook
    ("r3k2r/8/8/8/8/8/8/R2K3R w KQkq - 0 1", "kq"),          # white king off e1
    ("r3k2r/8/8/8/8/8/8/r3K2R w KQkq - 0 1", "Kkq"),         # the a1 rook is BLACK
    ("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", "KQkq"),        # nothing to clean
    ("4k2r/8/8/8/8/8/8/R3K3 w Qk - 0 1", "Qk"),
]


def _rights(state):
    return "".join(ch for ch, bit in (("K", 1), ("Q", 2), ("k", 4), ("q", 8)) if (state >> 1) & bit)


def test_fen_roots_carry_cleaned_castling_rights_and_repetition_counts_follow():
    """game.py:17-21 builds ``chess.Board(fen)``; python-chess reads castling rights only through
    ``clean_castling_rights()`` -- also in the transposition key of ``is_fivefold_repetition`` (game.py:92-109).
    With the FEN's letters taken as they stand a king's first move would clear bits python-chess never had and
    the root position would not count as a repetition of its later occurrences."""
    for fen, want in UNCLEAN_FENS:
        assert _rights(board_from_fen(fen).state) == want, fen
    g = OracleGame(board=board_from_fen(UNCLEAN_FENS[0][0]))
    for rep in range(4):
        assert g.get_result() is None
        for u in ["e1d1", "e8d8", "d1e1", "d8e8"]:
            assert g.move(u)
    # the root is the first of FIVE occurrences: python-chess ends the game here (16 plies), not 4 plies later
    assert g.repetitions() == 5 and g.get_result() == 0 and len(g) == 16
    # castling out of a cleaned root is not generated, out of an intact one it is
    assert "e1g1" not in OracleGame(board=board_from_fen(UNCLEAN_FENS[2][0])).get_legal_moves()
    assert "e1g1" in OracleGame(board=board_from_fen(UNCLEAN_FENS[4][0])).get_legal_moves()


def test_position_with_the_maximum_number_of_legal_moves():
    """218 legal moves (the published maximum): the size every move / edge array is built for."""
    from tests.util import MAX_MOVES_FEN
    g = OracleGame(board=board_from_fen(MAX_MOVES_FEN))
    lm = g.get_legal_moves()
    assert len(lm) == 218 and len(set(lm)) == 218
    assert g.get_result() is None


def test_python_chess_readme_known_answers():
    """The python-chess README's own worked examples (public known answers of the package the
    reference delegates to, requirements.txt:9): scholar's mate -- `board.is_checkmate()` True after
    e4 e5 Qh5 Nc6 Bc4 Nf6 Qxf7#; and, for that final position given as a FEN, `is_stalemate()`
    False, `is_insufficient_material()` False, `is_game_over()` True, `can_claim_fifty_moves()`
    False, `halfmove_clock` 0, `is_fivefold_repetition()` False, `is_seventyfive_moves()` False."""
    g = OracleGame()
    for u in ["e2e4", "e7e5", "d1h5", "b8c6", "f1c4", "g8f6"]:
        assert g.move(u) and g.get_result() is None
    assert g.move("h5f7")
    assert g.get_result() == 1 and g.get_legal_moves() == []            # checkmate, white won
    fen = "r1bqkb1r/pppp1Qpp/2n2n2/4p3/2B1P3/8/PPPP1PPP/RNB1K1NR b KQkq - 0 4"
    h = OracleGame(board=board_from_fen(fen))
    assert h.get_fen() == g.get_fen() == fen.split()[0]
    assert h.get_result() == 1 and h.repetitions() == 1
    assert (g.board_at(0).state >> 12) & 255 == 0                       # the capture reset the clock
    # README "Make and unmake moves": Nf3 is legal at the start, a8a1 is not
    s = OracleGame()
    assert "g1f3" in s.get_legal_moves() and not s.move("a8a1") and len(s.get_legal_moves()) == 20


def test_encoder_oracle_matches_the_reference_encoder(golden_dir):
    """encoder_cases.json: the reference's own netencoder.get_game_state (and helpers, netencoder.py:
    13-91), executed by oracle/make_golden.py over an adapter of the four python-chess Board /
    SquareSet members it touches.  Plane order, colour order, the per-colour 'empty' plane, the
    history stack (0, 1, 2, 7, 8, 9 ... 150 plies back, zeros where there is none; a FEN root with
    en passant and promotions) and the side-to-move plane."""
    import json
    import os
    import numpy as np
    from oracle import encoder_oracle
    cases = json.load(open(os.path.join(golden_dir, "encoder_cases.json")))["cases"]
    assert len(cases) >= 12
    for c in cases:
        g = OracleGame(board=board_from_fen(c["fen"])) if c["fen"] else OracleGame()
        for u in c["prefix_moves"]:
            assert g.move(u)
        ref = np.unpackbits(np.frombuffer(bytes.fromhex(c["planes_packbits_hex"]), np.uint8))[:8 * 8 * 127]
        mine = encoder_oracle.get_game_state(g)
        assert mine.shape == (8, 8, 127) and int(mine.sum()) == c["ones"]
        assert np.array_equal(mine.reshape(-1), ref.astype(np.float64)), (c["fen"], len(c["prefix_moves"]))
        assert bool(mine[0, 0, 126]) == c["turn"]
