"""CPU: the MCTS oracle against the committed golden vectors (outputs of the reference's own
mctree.py, oracle/make_golden.py) and, where /root/reference exists, against mctree.py live."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import mcts_oracle, ref_loader
from oracle.chess_oracle import OracleGame
from oracle.fakenet import FakeNet


def load_cases(golden_dir):
    with open(os.path.join(golden_dir, "mcts_cases.json")) as f:
        return json.load(f)["cases"]


def case_game(c):
    g = OracleGame()
    for u in c["prefix_moves"]:
        assert g.move(u)
    return g


def hexf64(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


def hexf32(h):
    return struct.unpack(">f", bytes.fromhex(h))[0]


def test_golden_has_a_case_where_numpy_modes_differ(golden_dir):
    cases = load_cases(golden_dir)
    assert any(c["differs_from_other_mode"] for c in cases)
    assert {c["mode"] for c in cases} == {"nep50", "legacy"}


def test_oracle_matches_golden_vectors(golden_dir):
    for c in load_cases(golden_dir):
        net = FakeNet(seed=c["net_seed"], prior_shift=c["prior_shift"], quant=c["quant"], tie=c["tie"])
        r = mcts_oracle.search(case_game(c), mcts_oracle.OracleAgent(net), c["sims"], noise=False,
                               mode=c["mode"])
        assert r.visits == c["visits"], c
        assert r.root_visits == c["root_visits"] == c["sims"] + 1
        assert r.moves == (c["bm"], c["am"])
        assert [struct.pack(">d", v).hex() for v in r.values] == c["values"]
        assert [struct.pack(">f", p).hex() for p in r.priors] == c["priors"]
        assert [struct.pack(">d", p).hex() for p in r.policy] == c["policy"]


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
