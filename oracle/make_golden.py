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
                      Round 2 adds the paths the random prefixes never reach: roots set up from
                      a FEN whose trees contain fifty-move claims, stalemates, mates after our
                      move (state = S1, mctree.py:241-246 skipped) and after the reply, terminal
                      nodes re-selected (mctree.py:218), a root whose chosen child ends the game
                      on our move (the ``move_stack[-2:]`` tuple, mctree.py:185-194, with and
                      without the IndexError branch), the 218-move position, a 120-ply quiet
                      game (halfmove clock > 100 plies deep into the ring) and 800-simulation
                      trees (the AlphaZero budget BASELINE.json's metric is quoted on).

* mcts_noise_cases.json -- ``search_move(noise=True)`` after ``np.random.seed(s)``: the noisy
                      policy bit for bit and the move it selects.
* selfplay_games.json -- whole games by the reference's own ``selfplay.play_game`` ->
                      ``AgentDistributed.best_move / predict_policy / predict_outcome`` ->
                      ``SelfPlayTree`` -> ``get_game_state`` (functions taken out of the parsed files by
                      oracle/ref_loader.py because their modules import TensorFlow / python-chess).
* encoder_cases.json -- ``netencoder.get_game_state`` (+ helpers) over an adapter of the four
                      python-chess Board / SquareSet members it uses.
* dataset_cases.json, sequence_cases.json -- ``dataset.DatasetGame`` (imported) and
                      ``netencoder.DataGameSequence`` (class statement executed) on two of those games.
* game_agent_cases.json -- ``gameagent.GameAgent`` (class statement executed) against a scripted human.
* model_graph.json -- the layers ``ChessModel.__init__`` / ``__res_block`` build, recorded by executing
                      them over stand-ins of the Keras constructors (topology only, no arithmetic).

What is NOT the reference's in these runs: the chess rules (the C oracle stands in for python-chess),
FakeNet (stands in for the Keras model's numbers) and the adapters named above.
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np

from . import mcts_oracle, ref_loader
from .chess_oracle import OracleGame, board_from_fen, move_to_uci
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


def quiet_game(seed, plies, min_clock=0):
    """A long game of mostly quiet piece moves: the halfmove clock climbs (to at least
    ``min_clock``: later draws are tried until one gets there) and the repetition scan has a long
    reversible tail to walk."""
    rng = np.random.default_rng(seed)
    while True:
        g = OracleGame()
        while len(g) < plies and g.get_result() is None:
            b = g.board_at(0)
            occ = 0
            for t in range(6):
                occ |= int(b.bb[t])
            lm = g.legal_move_ids()
            quiet = [m for m in lm if not (int(b.bb[0]) >> (m & 63)) & 1 and not (occ >> ((m >> 6) & 63)) & 1]
            pool = quiet if quiet and rng.random() < 0.97 else lm
            g.move(move_to_uci(pool[int(rng.integers(len(pool)))]))
        if g.get_result() is None and ((int(g.board_at(0).state) >> 12) & 255) >= min_clock:
            return g


MAX_MOVES_FEN = "R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1"

# Round-2 cases: dicts.  "fen" = root set up with an EMPTY move stack (python-chess Board(fen)),
# then "moves" pushed; "quiet" = quiet_game(seed, plies) from the standard position.
CASES2 = [
    dict(name="claim_clock96_w", fen="8/8/8/4k3/8/8/4K3/7R w - - 96 80", net=13, shift=30, sims=90),
    dict(name="claim_clock97_b", fen="8/8/8/4k3/8/8/4K3/7R b - - 97 80", net=13, shift=30, sims=90),
    dict(name="claim_clock99_b", fen="8/8/8/4k3/8/8/3QK3/8 b - - 99 90", net=14, shift=29, sims=60),
    dict(name="mate_or_stalemate_in_one", fen="7k/8/5KQ1/8/8/8/8/8 w - - 0 1", net=13, shift=30, sims=90),
    dict(name="back_rank_mate_no_stack", fen="6k1/5ppp/8/8/8/8/5PPP/R5K1 w - - 0 1", net=13, shift=30, sims=90),
    dict(name="back_rank_mate_tuple_quirk", fen="6k1/p4ppp/8/8/8/8/5PPP/1R4K1 b - - 0 1",
         moves=["a7a6"], net=15, shift=30, sims=120),
    dict(name="single_reply_black", fen="k7/8/1K6/8/8/8/8/2Q5 b - - 10 1", net=13, shift=30, sims=90),
    dict(name="castling_both_sides", fen="r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", net=13, shift=30, sims=90),
    dict(name="ep_and_promotions", fen="n1n5/PPPk4/8/8/4Pp2/8/4Kppp/5N1N b - e3 0 1", net=16, shift=28, sims=110),
    dict(name="max_moves_218", fen=MAX_MOVES_FEN, net=3, shift=30, sims=240),
    dict(name="quiet_120_plies", quiet=(77, 120), net=41, shift=31, sims=120),
    dict(name="quiet_170_plies", quiet=(78, 170), net=42, shift=29, sims=100),
    dict(name="quiet_clock_over_90", quiet=(79, 130, 92), net=43, shift=31, sims=160),
    dict(name="fivefold_on_our_move", fen="8/8/8/4k3/8/8/4K3/7R w - - 0 1",
         moves=["h1h2", "e5e6", "h2h1", "e6e5"] * 3 + ["h1h2", "e5e6", "h2h1"], net=21, shift=30, sims=60),
    dict(name="sims800_deep", prefix=(2, 7), net=4, shift=31, sims=800),
    dict(name="sims800_wide", prefix=(11, 40), net=19, shift=26, sims=800),
]

# Round-5 cases: FEN roots that CLAIM castling rights the position does not hold.  game.py:17-21 builds
# chess.Board(fen), which reads the rights only through clean_castling_rights() -- move generation, push and the
# transposition key of the fivefold rule (game.py:92-109).  In "no_rooks" the root is the FIRST of five occurrences:
# with the FEN's letters hashed as they stand the kings' first moves would change the key and d8e8 would only be
# the fourth.
_SHUFFLE_W = ["e1d1", "e8d8", "d1e1", "d8e8"]
_SHUFFLE_B = ["e8d8", "e1d1", "d8e8", "d1e1"]
CASES5 = [
    dict(name="unclean_rights_no_rooks_fivefold_on_our_move", fen="4k3/p7/8/8/8/8/P7/4K3 w KQkq - 0 1",
         moves=_SHUFFLE_W * 3 + _SHUFFLE_W[:3], net=21, shift=30, sims=60),
    dict(name="unclean_rights_one_rook_fivefold_on_our_move", fen="4k3/8/8/8/8/8/8/4K2R b KQkq - 0 1",
         moves=_SHUFFLE_B * 4 + _SHUFFLE_B[:1], net=22, shift=30, sims=60),
    dict(name="unclean_rights_king_off_e1", fen="r3k2r/8/8/8/8/8/8/R2K3R w KQkq - 0 1", net=13, shift=30, sims=90),
]


def write_r5(mct):
    cases5 = []
    for c in CASES5:
        a = run_case2(mct, c, "nep50")
        b = run_case2(mct, c, "legacy")
        a["differs_from_other_mode"] = b["differs_from_other_mode"] = a["visits"] != b["visits"]
        cases5 += [a, b]
        print(c["name"], "children", len(a["visits"]), "nodes", a["n_nodes"], "terminal nodes", a["n_terminal_nodes"],
              "moves", (a["bm"], a["am"]), "chosen child result", a["chosen_child_result"])
    with open(os.path.join(OUT, "mcts_cases_r5.json"), "w") as f:
        json.dump({"source": "mctree.SelfPlayTree.search_move (mctree.py:159-198) imported from /root/reference with a "
                             "stub game module; numpy %s; FEN roots claiming castling rights the position does not hold "
                             "(python-chess clean_castling_rights, restated in oracle/chess_oracle.py:board_from_fen)"
                             % np.__version__,
                   "cases": cases5}, f)


def case2_game(c):
    if "fen" in c:
        g = OracleGame(board=board_from_fen(c["fen"]))
    elif "quiet" in c:
        g = quiet_game(*c["quiet"])
    else:
        g = prefix_game(*c["prefix"])
    for u in c.get("moves", []):
        assert g.move(u), (c["name"], u)
    return g


def run_case2(mct, c, mode):
    g = case2_game(c)
    net = FakeNet(seed=c["net"], prior_shift=c["shift"], quant=c.get("quant", 0))
    agent = mcts_oracle.OracleAgent(net, widen_priors=(mode == "legacy"))
    tree = mct.SelfPlayTree(g, threads=1)
    bm, am = tree.search_move(agent, max_iters=c["sims"], noise=False, ai_move=True)
    kids = tree.root.children
    pol = tree.compute_policy(tree.root, noise=False)
    chosen = int(np.argmax(pol)) if len(kids) else -1

    def count(n, pred):
        return int(pred(n)) + sum(count(k, pred) for k in n.children)

    return {
        "name": c["name"], "fen": c.get("fen"), "mode": mode,
        "net_seed": c["net"], "prior_shift": c["shift"], "quant": c.get("quant", 0), "tie": False,
        "sims": c["sims"],
        # moves on the root's move stack (pushed after "fen", or from the standard position)
        "prefix_moves": [m.uci() for m in g.board.move_stack],
        "root_result": g.get_result(),
        "visits": [int(k.visits) for k in kids],
        "values": [f64hex(k.value) for k in kids],
        "priors": [f32hex(k.prior) for k in kids],
        "root_visits": int(tree.root.visits),
        "bm": bm, "am": am, "chosen": chosen,
        "chosen_child_result": kids[chosen].state.get_result() if chosen >= 0 else None,
        "chosen_child_stack": len(kids[chosen].state.board.move_stack) if chosen >= 0 else 0,
        "policy": [f64hex(p) for p in pol],
        "n_evals": agent.n_evals,
        "n_nodes": count(tree.root, lambda n: True),
        "n_terminal_nodes": count(tree.root, lambda n: n.state.get_result() is not None),
        "terminal_visits": sum_terminal_visits(tree.root),
    }


def run_noise_case(mct, c, seed):
    """search_move(noise=True) of the reference with its global np.random seeded: the move it
    returns and the noisy policy it took the argmax of (mctree.py:177, 317-321: tau from the
    root's ply count, 0.75 x policy + an UNSCALED Dirichlet(0.03) draw)."""
    g = case2_game(c)
    net = FakeNet(seed=c["net"], prior_shift=c["shift"], quant=c.get("quant", 0))
    agent = mcts_oracle.OracleAgent(net)
    tree = mct.SelfPlayTree(g, threads=1)
    np.random.seed(seed)
    bm, am = tree.search_move(agent, max_iters=c["sims"], noise=True, ai_move=True)
    kids = tree.root.children
    np.random.seed(seed)                                # the same draw search_move just made
    pol = tree.compute_policy(tree.root, noise=True)
    chosen = int(np.argmax(pol)) if len(kids) else -1
    quiet = tree.compute_policy(tree.root, noise=False)
    return {
        "name": c["name"], "fen": c.get("fen"), "mode": "nep50", "noise_seed": seed,
        "net_seed": c["net"], "prior_shift": c["shift"], "quant": c.get("quant", 0), "sims": c["sims"],
        "prefix_moves": [m.uci() for m in g.board.move_stack],
        "root_plies": len(g.board.move_stack),
        "visits": [int(k.visits) for k in kids], "root_visits": int(tree.root.visits),
        "bm": bm, "am": am, "chosen": chosen,
        "chosen_without_noise": int(np.argmax(quiet)) if len(kids) else -1,
        "policy_noise": [f64hex(p) for p in pol],
    }


GAMES = [  # (seed of random / np.random, FakeNet seed, prior shift, simulations per move)
    (1, 5, 26, 40), (2, 6, 28, 40), (3, 7, 24, 25), (4, 8, 27, 40), (7, 9, 26, 30), (8, 10, 29, 40),
]


def run_game(mct, seed, net_seed, shift, sims):
    """One whole game by the reference's own play_game -> AgentDistributed.best_move -> SelfPlayTree
    (noise on, global random / np.random seeded).  play_game asks for max_iters=900 (selfplay.py:76);
    the agent here is a stub anyway, and its best_move runs the reference's best_move with `sims`
    simulations instead so that the fixture can be replayed in seconds (ref_loader.make_reference_agent)."""
    import random
    play_game, _ = ref_loader.load_play_game(mct)
    # best_move, predict_policy, predict_outcome, predict: the reference's; the network round trip
    # encodes with the reference's get_game_state and evaluates FakeNet
    agent = ref_loader.make_reference_agent(mct, FakeNet(seed=net_seed, prior_shift=shift), sims)
    random.seed(seed)
    np.random.seed(seed)
    g = play_game(agent)
    return {"seed": seed, "net_seed": net_seed, "prior_shift": shift, "sims": sims,
            "player_color": bool(g.player_color), "moves": [m.uci() for m in g.board.move_stack],
            "result": g.get_result(), "n_evals": agent.n_evals}


def dataset_case(ds_mod, gm, date):
    """dataset.DatasetGame of the reference (dataset.py:7-97) on one of the golden games: the JSON
    text it serialises, what loads() makes of that text, and augment_game's expansion."""
    g = OracleGame(player_color=gm["player_color"], date=date)
    for u in gm["moves"]:
        assert g.move(u)
    ds = ds_mod.DatasetGame()
    ds.append(g)
    text = str(ds)
    back = ds_mod.DatasetGame()
    back.loads(text)
    assert len(back) == 1 and str(back) == text
    aug = ds.augment_game(g)
    return {"seed": gm["seed"], "date": date, "player_color": gm["player_color"], "moves": gm["moves"],
            "json": text,
            "augment": [{"plies": len(a["game"]), "fen": a["game"].get_fen(), "next_move": a["next_move"],
                         "result": a["result"]} for a in aug]}


def sequence_case(ds_mod, seq_cls, dsc, seed, flips):
    """One batch of the reference's DataGameSequence over the two dataset_cases games."""
    ds = ds_mod.DatasetGame()
    ds.loads("[" + ", ".join(c["json"][1:-1] for c in dsc) + "]")
    seq = seq_cls(ds, batch_size=2, random_flips=flips)
    assert len(seq) == 1
    np.random.seed(seed)
    x, (pol, val) = seq[0]
    assert x.shape[1:] == (8, 8, 127) and pol.shape == (x.shape[0], 1968) and pol.dtype == np.float32
    assert (pol.sum(1) == 1).all()
    return {"seed": seed, "random_flips": flips, "n": int(x.shape[0]), "x_dtype": str(x.dtype),
            "x_sha256": hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest(),
            "first_sample_of_each_game_packbits_hex": [
                np.packbits(x[i].astype(np.uint8).reshape(-1)).tobytes().hex() for i in (0, len(dsc[0]["moves"]))],
            "labels": [int(i) for i in pol.argmax(1)], "values": [int(v) for v in val]}


def game_agent_case(mct, agent_white, net_seed, human_seed):
    """The reference's own GameAgent (gameagent.py:7-50) against an agent made of the reference's
    best_move / predict_policy (ref_loader.make_reference_agent): a scripted human plays seeded random
    legal moves, tries an illegal move before each of them, and (white agent) a first call whose
    argument is ignored.  Every call: argument, return value, plies afterwards."""
    GameAgent, AgentBase = ref_loader.load_game_agent()
    inner = ref_loader.make_reference_agent(mct, FakeNet(seed=net_seed, prior_shift=30), 1)

    class ScriptAgent(AgentBase):
        color = agent_white

        def best_move(self, game, real_game=False, **kw):
            return inner.best_move(game, real_game=real_game, **kw)

    g = GameAgent(ScriptAgent(), player_color=not agent_white)
    rng = np.random.default_rng(human_seed)
    calls = []

    def call(mv):
        ok = g.move(mv)
        calls.append({"move": mv, "returned": bool(ok), "plies": len(g)})

    if agent_white:
        call("a7a6")
    for _ in range(40):
        if g.get_result() is not None:
            break
        call("a1a1")
        lm = g.get_legal_moves()
        call(lm[int(rng.integers(len(lm)))])
    copy = g.get_copy()
    return {"agent_white": agent_white, "net_seed": net_seed, "prior_shift": 30, "human_seed": human_seed, "calls": calls,
            "moves": g.get_history()["moves"], "result": g.get_result(),
            "copy_is_game_agent": type(copy).__name__ == "GameAgent", "copy_moves": copy.get_history()["moves"]}


ENCODER_CASES = [  # (prefix seed, plies) from the standard position, or a FEN root + pushed moves
    (1, 0), (2, 1), (3, 2), (4, 7), (5, 8), (6, 9), (7, 15), (8, 40), (9, 91), (10, 150),
    {"fen": "r3k2r/pPp2ppp/8/3pP3/8/8/P1P2PpP/R3K2R w KQkq d6", "moves": ["e5d6", "g2h1n", "b7a8q"]},
    {"fen": "8/8/8/4k3/8/8/4K3/7R b", "moves": []},
]


def encoder_case(ref_encode, spec):
    if isinstance(spec, dict):
        g = OracleGame(board=board_from_fen(spec["fen"]))
        for u in spec["moves"]:
            assert g.move(u), u
        fen, pre = spec["fen"], list(spec["moves"])
    else:
        g = prefix_game(*spec)
        fen, pre = None, [m.uci() for m in g.board.move_stack]
    planes = np.asarray(ref_encode(g))
    assert planes.shape == (8, 8, 127) and set(np.unique(planes)) <= {0.0, 1.0}
    return {"fen": fen, "prefix_moves": pre, "turn": bool(g.turn),
            # (8, 8, 127) of 0/1 -> np.packbits over the flattened [row][col][channel] order
            "planes_packbits_hex": np.packbits(planes.astype(np.uint8).reshape(-1)).tobytes().hex(),
            "ones": int(planes.sum())}


def sum_terminal_visits(n):
    own = n.visits if n.state.get_result() is not None else 0
    return int(own) + sum(sum_terminal_visits(k) for k in n.children)


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
    if sys.argv[1:] == ["--only", "r5"]:         # just the round-5 fixture (the others take ~4.5 minutes)
        write_r5(ref_loader.load_mctree())
        return
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
    cases2 = []
    for c in CASES2:
        a = run_case2(mct, c, "nep50")
        b = run_case2(mct, c, "legacy")
        a["differs_from_other_mode"] = b["differs_from_other_mode"] = a["visits"] != b["visits"]
        cases2 += [a, b]
        print(c["name"], "children", len(a["visits"]), "nodes", a["n_nodes"], "terminal nodes",
              a["n_terminal_nodes"], "terminal visits", a["terminal_visits"], "moves", (a["bm"], a["am"]),
              "chosen child result", a["chosen_child_result"], "modes differ", a["visits"] != b["visits"])
    with open(os.path.join(OUT, "mcts_cases_r2.json"), "w") as f:
        json.dump({"source": "mctree.SelfPlayTree.search_move (mctree.py:159-198) imported from "
                             "/root/reference with a stub game module; numpy %s; roots from FENs / long "
                             "quiet games / 800 simulations" % np.__version__,
                   "cases": cases2}, f)
    write_r5(mct)
    noisy = []
    for i, c in enumerate(CASES2):
        if c["sims"] > 200 or not case2_game(c).get_legal_moves():
            continue
        for rep in range(2):
            a = run_noise_case(mct, c, 1000 + 17 * i + rep)
            noisy.append(a)
        print(c["name"], "noise moves", (a["bm"], a["am"]), "argmax", a["chosen"], "without noise", a["chosen_without_noise"])
    with open(os.path.join(OUT, "mcts_noise_cases.json"), "w") as f:
        json.dump({"source": "mctree.SelfPlayTree.search_move(noise=True) (mctree.py:159-198, 305-322) imported "
                             "from /root/reference, np.random.seed(noise_seed) before the call; numpy %s"
                             % np.__version__, "cases": noisy}, f)
    ref_encode = ref_loader.load_encoder()
    enc = [encoder_case(ref_encode, spec) for spec in ENCODER_CASES]
    with open(os.path.join(OUT, "encoder_cases.json"), "w") as f:
        json.dump({"source": "netencoder.get_game_state and helpers (netencoder.py:13-91), taken out of the parsed "
                             "reference file and executed over oracle/ref_loader.py's adapter of the four "
                             "python-chess Board / SquareSet members they use (pieces, mirror, tolist, pop); the "
                             "positions come from the C oracle", "cases": enc}, f)
    print("encoder cases", len(enc), "ones per case", [e["ones"] for e in enc])
    games = []
    for spec in GAMES:
        gm = run_game(mct, *spec)
        games.append(gm)
        print("game", spec, "agent is white:", gm["player_color"], "plies", len(gm["moves"]), "result", gm["result"])
    with open(os.path.join(OUT, "selfplay_games.json"), "w") as f:
        json.dump({"source": "selfplay.play_game (selfplay.py:59-84) -> AgentDistributed.best_move "
                             "(agentdistributed.py:39-68) -> mctree.SelfPlayTree, the three executed from "
                             "/root/reference (oracle/ref_loader.py); random.seed(seed), np.random.seed(seed); "
                             "`sims` simulations per move instead of play_game's 900; numpy %s" % np.__version__,
                   "games": games}, f)
    ds_mod = ref_loader.load_dataset()
    dsc = [dataset_case(ds_mod, gm, "01/02/2020 03:04:05") for gm in sorted(games, key=lambda x: len(x["moves"]))[:2]]
    with open(os.path.join(OUT, "dataset_cases.json"), "w") as f:
        json.dump({"source": "dataset.DatasetGame (dataset.py:7-97) imported from /root/reference with the stub game "
                             "module, on two of the games of selfplay_games.json", "cases": dsc}, f)
    print("dataset cases", [(c["seed"], len(c["augment"])) for c in dsc])
    seq_cls = ref_loader.load_data_sequence(ds_mod)
    seqs = [sequence_case(ds_mod, seq_cls, dsc, seed, flips) for seed, flips in ((3, 0.0), (1, 0.5), (6, 0.5), (2, 0.5))]
    with open(os.path.join(OUT, "sequence_cases.json"), "w") as f:
        json.dump({"source": "netencoder.DataGameSequence (netencoder.py:137-181) taken out of the parsed reference "
                             "file and executed (oracle/ref_loader.py) over the reference's dataset.DatasetGame of "
                             "the two games of dataset_cases.json; batch_size 2, np.random.seed(seed) before seq[0]",
                   "cases": seqs}, f)
    print("sequence cases", [(c["seed"], c["n"], c["x_sha256"][:8]) for c in seqs])
    gac = [game_agent_case(mct, aw, ns_, hs) for aw, ns_, hs in ((True, 41, 12), (False, 41, 12), (True, 42, 13), (False, 43, 14))]
    with open(os.path.join(OUT, "game_agent_cases.json"), "w") as f:
        json.dump({"source": "gameagent.GameAgent (gameagent.py:7-50), its class statement executed from /root/reference "
                             "(oracle/ref_loader.load_game_agent) against an agent made of the reference's best_move / "
                             "predict_policy and FakeNet", "cases": gac}, f)
    print("game agent cases", [(c["agent_white"], len(c["calls"]), len(c["moves"])) for c in gac])
    graph = ref_loader.record_model_graph()
    with open(os.path.join(OUT, "model_graph.json"), "w") as f:
        json.dump({"source": "ChessModel.__init__ and __res_block (model.py:17-72,111-122) executed from /root/reference "
                             "over recording stand-ins of the Keras constructors (oracle/ref_loader.record_model_graph): "
                             "the layers the reference builds, their arguments and wiring, and the compile() call",
                   "graph": graph}, f)
    print("model graph", len(graph["nodes"]), "nodes")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
