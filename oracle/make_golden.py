"""oracle/make_golden.py -- TEST INFRASTRUCTURE: writes tests/golden/*.json.

Run in the authoring container only (needs /root/reference):

    python -m oracle.make_golden

* uci_labels.json  -- output of the reference's own ``netencoder.get_uci_labels``
                      (netencoder.py:94-134, exec'd in place) + its sha256.
* mcts_cases.json  -- outputs of the reference's own ``mctree.SelfPlayTree.search_move``
                      (mctree.py:159-198; threads=1, noise off) run on the C-oracle chess
                      rules with the deterministic FakeNet: root children visit counts,
                      value sums (float64 hex), priors (float32 hex), chosen (bm, am),
                      ``compute_policy`` output.  Two numpy promotion modes: "nep50" is the
                      reference code as it runs under this image's numpy 2.x; "legacy" feeds
                      it float64-widened priors, which makes ``10 * prior`` a float64 product
                      exactly as numpy 1.17.2 (requirements.txt:6) would compute it.

Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import hashlib
import json
import os
import struct

import numpy as np

from . import mcts_oracle, ref_loader
from .chess_oracle import OracleGame, move_to_uci
from .fakenet import FakeNet

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def f64hex(x):
    return struct.pack(">d", float(x)).hex()


def f32hex(x):
    return struct.pack(">f", float(x)).hex()


def prefix_game(seed, plies):
    rng = np.random.default_rng(seed)
    g = OracleGame()
    while len(g) < plies and g.get_result() is None:
        lm = g.legal_move_ids()
        g.move(move_to_uci(lm[int(rng.integers(len(lm)))]))
    return g


CASES = [
    # (prefix seed, plies, net seed, prior_shift, quant, sims)
    (1, 0, 3, 29, 0, 60),
    (2, 7, 4, 24, 0, 80),
    (3, 16, 5, 31, 0, 120),
    (4, 31, 6, 33, 12, 120),
    (5, 44, 7, 30, 16, 100),
    (6, 60, 8, 29, 0, 25),
    (7, 12, 9, 36, 18, 150),
    (8, 23, 10, 27, 8, 90),
    # FakeNet(tie=True): adjacent priors that tie only when 10*prior is rounded to float32
    (9, 10, 11, 30, 0, 140, True),
    (10, 28, 12, 31, 0, 140, True),
]


def run_case(mct, case, mode):
    pseed, plies, nseed, shift, quant, sims = case[:6]
    tie = len(case) > 6 and case[6]
    g = prefix_game(pseed, plies)
    net = FakeNet(seed=nseed, prior_shift=shift, quant=quant, tie=tie)
    agent = mcts_oracle.OracleAgent(net, widen_priors=(mode == "legacy"))
    tree = mct.SelfPlayTree(g, threads=1)
    bm, am = tree.search_move(agent, max_iters=sims, noise=False, ai_move=True)
    kids = tree.root.children
    pol = tree.compute_policy(tree.root, noise=False)
    return {
        "prefix_seed": pseed, "plies": plies, "net_seed": nseed, "prior_shift": shift,
        "quant": quant, "sims": sims, "mode": mode, "tie": bool(tie),
        "prefix_moves": [m.uci() for m in g.board.move_stack],
        "visits": [int(c.visits) for c in kids],
        "values": [f64hex(c.value) for c in kids],
        "priors": [f32hex(c.prior) for c in kids],
        "root_visits": int(tree.root.visits),
        "bm": bm, "am": am,
        "policy": [f64hex(p) for p in pol],
        "n_evals": agent.n_evals,
    }


def main():
    os.makedirs(OUT, exist_ok=True)
    labels = ref_loader.load_uci_labels()
    with open(os.path.join(OUT, "uci_labels.json"), "w") as f:
        json.dump({"source": "netencoder.get_uci_labels (netencoder.py:94-134) exec'd from /root/reference",
                   "sha256": hashlib.sha256("\n".join(labels).encode()).hexdigest(),
                   "labels": labels}, f)
    mct = ref_loader.load_mctree()
    cases = []
    for case in CASES:
        a = run_case(mct, case, "nep50")
        b = run_case(mct, case, "legacy")
        a["differs_from_other_mode"] = b["differs_from_other_mode"] = a["visits"] != b["visits"]
        cases += [a, b]
        print(case, "modes differ:", a["visits"] != b["visits"], "children:", len(a["visits"]))
    with open(os.path.join(OUT, "mcts_cases.json"), "w") as f:
        json.dump({"source": "mctree.SelfPlayTree.search_move (mctree.py:159-198) imported from "
                             "/root/reference with a stub game module; numpy %s" % np.__version__,
                   "cases": cases}, f)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
