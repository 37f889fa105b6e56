"""CPU: the MCTS oracle against the committed golden vectors (outputs of the reference's own
mctree.py, oracle/make_golden.py) and, where /root/reference exists, against mctree.py live."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import mcts_oracle, ref_loader
from oracle.chess_oracle import OracleGame, board_from_fen
from oracle.fakenet import FakeNet


def load_cases(golden_dir, name="mcts_cases.json"):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)["cases"]


def case_game(c):
    g = OracleGame(board=board_from_fen(c["fen"])) if c.get("fen") else OracleGame()
    for u in c["prefix_moves"]:
        assert g.move(u)
    return g


def test_noisy_policy_and_move_match_the_reference_with_a_seeded_global_stream(golden_dir):
    """mcts_noise_cases.json: the reference's search_move(noise=True) after np.random.seed(s).  The
    oracle's search (drawing from a RandomState(s), the generator behind the seeded global stream)
    returns the same moves; the oracle's and the product's host compute_policy (chessrl_amd/engine.py:
    pure numpy, no GPU) give the same noisy policy bit for bit, and choose_children the same argmax --
    incl. tau < 1 roots (>= 30 plies) and cases where the noise changes the move."""
    import numpy as np
    from chessrl_amd import engine
    cases = load_cases(golden_dir, "mcts_noise_cases.json")
    assert any(c["chosen"] != c["chosen_without_noise"] for c in cases)
    assert any(c["root_plies"] >= 30 for c in cases) and any(c["root_plies"] < 30 for c in cases)
    for c in cases:
        net = FakeNet(seed=c["net_seed"], prior_shift=c["prior_shift"], quant=c["quant"])
        r = mcts_oracle.search(case_game(c), mcts_oracle.OracleAgent(net), c["sims"], noise=True,
                               mode="nep50", rng=np.random.RandomState(c["noise_seed"]))
        assert r.visits == c["visits"] and r.moves == (c["bm"], c["am"]) and r.chosen == c["chosen"], c["name"]
        for fn in (mcts_oracle.compute_policy, engine.compute_policy):
            pol = fn(c["visits"], c["root_visits"], c["root_plies"], noise=True,
                     rng=np.random.RandomState(c["noise_seed"]))
            assert [struct.pack(">d", p).hex() for p in pol] == c["policy_noise"], (c["name"], fn.__module__)
        vis = np.zeros((1, 256), np.int64)
        vis[0, :len(c["visits"])] = c["visits"]
        got = engine.choose_children(vis, [len(c["visits"])], [c["root_visits"]], [c["root_plies"]], noise=True,
                                     rngs=[np.random.RandomState(c["noise_seed"])])
        assert got[0] == c["chosen"], c["name"]


def test_whole_games_match_the_reference_play_game(golden_dir):
    """selfplay_games.json: whole games played by the reference's OWN selfplay.play_game ->
    AgentDistributed.best_move -> mctree.SelfPlayTree (executed from /root/reference by
    oracle/make_golden.py; noise on, random / np.random seeded).  The oracle's restatement of that
    loop replays them move for move: colour draw, the opponent's greedy opening when the agent is
    black, (our move, stored reply) per search, termination and result.  (Three of the six games
    here; the GPU test replays all six through the drop-in objects.)"""
    import random
    import numpy as np
    games = json.load(open(os.path.join(golden_dir, "selfplay_games.json")))["games"]
    assert {g["player_color"] for g in games} == {True, False} and len(games) >= 6
    picked = sorted(games, key=lambda g: len(g["moves"]) * g["sims"])[:2] + [g for g in games if g["player_color"]][:1]
    for gm in picked:
        agent = mcts_oracle.OracleAgent(FakeNet(seed=gm["net_seed"], prior_shift=gm["prior_shift"]))
        random.seed(gm["seed"])
        np.random.seed(gm["seed"])
        og = mcts_oracle.play_game(agent, max_iters=gm["sims"], noise=True)
        assert og.get_history()["moves"] == gm["moves"], gm["seed"]
        assert og.get_result() == gm["result"] and bool(og.player_color) == gm["player_color"]
        assert agent.n_evals == gm["n_evals"]


def hexf64(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


def hexf32(h):
    return struct.unpack(">f", bytes.fromhex(h))[0]


def test_golden_has_a_case_where_numpy_modes_differ(golden_dir):
    cases = load_cases(golden_dir)
    assert any(c["differs_from_other_mode"] for c in cases)
    assert {c["mode"] for c in cases} == {"nep50", "legacy"}


@pytest.mark.parametrize("fixture", ["mcts_cases.json", "mcts_cases_r2.json", "mcts_cases_r5.json"])
def test_oracle_matches_golden_vectors(golden_dir, fixture):
    """mcts_cases_r2.json: roots from FENs (fifty-move claims, mates and stalemates on our move and
    after the reply, fivefold repetition reached by our move, the (previous ply, our move) tuple of
    mctree.py:185-194 with and without its IndexError branch), the 218-move root, 120-170-ply quiet
    games and 800-simulation trees -- all outputs of the reference's own mctree.py."""
    for c in load_cases(golden_dir, fixture):
        net = FakeNet(seed=c["net_seed"], prior_shift=c["prior_shift"], quant=c["quant"], tie=c["tie"])
        r = mcts_oracle.search(case_game(c), mcts_oracle.OracleAgent(net), c["sims"], noise=False,
                               mode=c["mode"])
        assert r.visits == c["visits"], c
        assert r.root_visits == c["root_visits"] == c["sims"] + 1
        assert r.moves == (c["bm"], c["am"])
        assert [struct.pack(">d", v).hex() for v in r.values] == c["values"]
        assert [struct.pack(">f", p).hex() for p in r.priors] == c["priors"]
        assert [struct.pack(">d", p).hex() for p in r.policy] == c["policy"]
        if "n_nodes" in c:
            assert r.n_nodes == c["n_nodes"] and r.chosen == c["chosen"]


def test_round2_goldens_reach_the_terminal_paths(golden_dir):
    """The fixture must keep exercising what it was made for (mctree.py:216-229,241-246,266-268,
    185-194): terminal nodes re-selected, games ending on our move, both tuple-quirk shapes, the
    218-child root and the 800-simulation budget."""
    cs = {c["name"]: c for c in load_cases(golden_dir, "mcts_cases_r2.json") if c["mode"] == "nep50"}
    assert sum(c["terminal_visits"] > c["n_terminal_nodes"] for c in cs.values()) >= 4   # re-selected
    q = cs["back_rank_mate_tuple_quirk"]
    assert q["chosen_child_result"] == 1 and (q["bm"], q["am"]) == ("a7a6", "b1b8")
    assert q["chosen_child_stack"] == 2
    n = cs["back_rank_mate_no_stack"]
    assert n["chosen_child_result"] == 1 and (n["bm"], n["am"]) == ("00000", "00000")
    f = cs["fivefold_on_our_move"]
    assert f["chosen_child_result"] == 0 and (f["bm"], f["am"]) == ("h2h1", "e6e5")
    assert len(cs["max_moves_218"]["visits"]) == 218
    assert cs["quiet_clock_over_90"]["n_terminal_nodes"] > 0
    assert sorted(c["sims"] for c in cs.values())[-2:] == [800, 800]


def test_round5_goldens_hold_the_fivefold_of_a_cleaned_fen_root(golden_dir):
    """mcts_cases_r5.json (FEN roots claiming castling rights the position does not hold): each shuffle tree holds
    exactly one terminal node -- the move that repeats a position for the fifth time, the ROOT position counted as
    python-chess counts it (clean_castling_rights in the transposition key); with the FEN's letters hashed as they
    stand "no_rooks" would hold none."""
    cs = {c["name"]: c for c in load_cases(golden_dir, "mcts_cases_r5.json") if c["mode"] == "nep50"}
    assert cs["unclean_rights_no_rooks_fivefold_on_our_move"]["n_terminal_nodes"] == 1
    assert cs["unclean_rights_one_rook_fivefold_on_our_move"]["n_terminal_nodes"] == 1
    assert len(cs["unclean_rights_king_off_e1"]["visits"]) == 24      # no white castling among the root's moves


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present on this box")
@pytest.mark.parametrize("mode", ["nep50", "legacy"])
def test_oracle_matches_reference_mctree_live(mode):
    mct = ref_loader.load_mctree()
    g = OracleGame()
    for u in ["d2d4", "g8f6", "c2c4", "e7e6", "b1c3"]:
        g.move(u)
    net = FakeNet(seed=17, prior_shift=30)
    agent = mcts_oracle.OracleAgent(net, widen_priors=(mode == "legacy"))
    tree = mct.SelfPlayTree(g, threads=1)
    mv = tree.search_move(agent, max_iters=70, noise=False, ai_move=True)
    r = mcts_oracle.search(g, mcts_oracle.OracleAgent(net), 70, noise=False, mode=mode)
    assert [c.visits for c in tree.root.children] == r.visits and mv == r.moves
    assert [float(c.value) for c in tree.root.children] == r.values
    np.random.seed(3)
    p_ref = tree.compute_policy(tree.root, noise=True)
    np.random.seed(3)
    p_mine = mcts_oracle.compute_policy(r.visits, r.root_visits, len(g), noise=True)
    assert np.array_equal(p_ref, p_mine)


def test_tau_schedule_and_denominator():
    """mctree.py:305-316: tau = 1 below 30 plies, nb/(1+nb^1.3) after; denominator root.visits."""
    p = mcts_oracle.compute_policy([3, 1], 5, 10, noise=False)
    assert np.array_equal(p, np.array([3 / 5, 1 / 5]))
    nb = 40
    tau = nb / (1 + np.power(nb, 1.3))
    p = mcts_oracle.compute_policy([3, 1], 5, nb, noise=False)
    assert np.array_equal(p, np.array([np.power(3, 1 / tau), np.power(1, 1 / tau)]) / np.power(5, 1 / tau))
