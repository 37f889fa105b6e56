"""oracle/pin_external.py -- TEST INFRASTRUCTURE: pins what this image cannot pin today.

The rules half (move ORDER, Game.get_result), the encoder and the tower of the reference live in
third-party packages that are absent here: python-chess==0.28.3 (requirements.txt:9) and
TensorFlow/Keras (model.py:6).  Run this ONE command on any machine where they import, next to a
checkout of the reference:

    python -m oracle.pin_external [--reference /root/reference]

and it regenerates, from the reference's own code run on the real packages,

    tests/golden/external_rules.json    legal-move lists in python-chess order, Game.get_result(),
                                        halfmove clock / ep / castling state after every ply of
                                        seeded random games + the perft and rule-corner positions
    tests/golden/external_encoder.json  netencoder.get_game_state planes (as set-bit lists) on
                                        positions with history                (needs chess)
    tests/golden/external_tower.npz     ChessModel.predict outputs + the Keras weights that made
                                        them, for a 2-block/32-filter tower   (needs tensorflow)

tests/test_external_pins.py consumes whichever of these files exist (CPU: the oracle; GPU: the HIP
path) and skips the rest.  Nothing here is imported by the product.  Fixtures are data; no
reference source text is stored.
"""
import argparse
import importlib
import json
import os
import random
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

RULE_FENS = [
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
    "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1",
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8",
    "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10",
    "8/8/8/KPp4r/8/8/8/4k3 w - c6 0 2", "8/8/8/2k5/3Pp3/8/8/4K3 b - d3 0 1",
    "n1n5/PPPk4/8/8/8/8/4Kppp/5N1N b - - 0 1", "4k3/8/8/8/8/5n2/4r3/4K3 w - - 0 1",
    "r3k2r/1P6/8/3pP3/8/8/P7/R3K2R w KQkq d6 0 2", "4k3/8/8/8/7b/8/3N4/R3K3 w Q - 0 1",
    "R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1",
    "8/8/8/4k3/8/8/4K3/7R w - - 99 1", "8/8/8/4k3/8/8/4K3/7R w - - 100 1",
    "8/8/8/4k3/8/8/4K3/7R b - - 99 80", "7k/6Q1/6K1/8/8/8/8/8 b - - 100 1",
    "8/8/8/4k3/8/8/4K3/7B w - - 0 1", "8/8/8/4k3/8/8/4K3/6NN w - - 0 1",
    "8/8/4b3/4k3/8/8/4K3/5B2 w - - 0 1", "8/8/4b3/4k3/8/8/4K3/4B3 w - - 0 1",
    "7k/5Q2/6K1/8/8/8/8/8 b - - 0 1", "8/8/8/4k3/8/8/4K3/7R w - - 149 1", "8/8/8/4k3/8/8/4K3/7R w - - 150 1",
]


def _state(board):
    return {"turn": bool(board.turn), "castling": board.castling_xfen(), "ep": board.ep_square,
            "has_legal_ep": bool(board.has_legal_en_passant()), "clock": board.halfmove_clock,
            "board_fen": board.board_fen()}


def pin_rules(ref_dir):
    import chess
    sys.path.insert(0, os.path.join(ref_dir, "src", "chessrl"))
    game_mod = importlib.import_module("game")             # the reference's own Game wrapper
    out = {"python_chess": chess.__version__, "positions": [], "games": []}
    for fen in RULE_FENS:
        g = game_mod.Game(board=chess.Board(fen))
        out["positions"].append({"fen": fen, "legal": g.get_legal_moves(), "result": g.get_result(),
                                 "state": _state(g.board)})
    for seed in range(24):                                  # seeded random games, every ply
        rng = random.Random(seed)
        g = game_mod.Game()
        plies = []
        while g.get_result() is None and len(g) < 400:
            legal = g.get_legal_moves()
            mv = legal[rng.randrange(len(legal))]
            plies.append({"legal": legal, "move": mv})
            assert g.move(mv)
            plies[-1]["result_after"] = g.get_result()
            plies[-1]["state_after"] = _state(g.board)
        out["games"].append({"seed": seed, "plies": plies})
    # a fivefold repetition and the threefold that must NOT end the game (game.py:94-96)
    g = game_mod.Game()
    seq = []
    for _ in range(4):
        for mv in ["g1f3", "g8f6", "f3g1", "f6g8"]:
            g.move(mv)
            seq.append({"move": mv, "result_after": g.get_result()})
    out["repetition"] = seq
    with open(os.path.join(OUT, "external_rules.json"), "w") as f:
        json.dump(out, f)
    print("wrote external_rules.json (python-chess %s)" % chess.__version__)


def pin_encoder(ref_dir):
    import chess                                            # noqa: F401
    import numpy as np
    sys.path.insert(0, os.path.join(ref_dir, "src", "chessrl"))
    netencoder = importlib.import_module("netencoder")      # imports tensorflow at module level
    game_mod = importlib.import_module("game")
    cases = []
    for seed, plies in [(0, 0), (1, 1), (2, 7), (3, 12), (4, 40), (5, 90)]:
        rng = random.Random(seed)
        g = game_mod.Game()
        while len(g) < plies and g.get_result() is None:
            legal = g.get_legal_moves()
            g.move(legal[rng.randrange(len(legal))])
        for flipped in (False, True):
            planes = np.asarray(netencoder.get_game_state(g, flipped=flipped))
            cases.append({"moves": g.get_history()["moves"], "flipped": flipped, "shape": list(planes.shape),
                          "ones": np.flatnonzero(planes.reshape(-1) != 0).tolist(),
                          "all_binary": bool(np.isin(planes, (0, 1)).all())})
    with open(os.path.join(OUT, "external_encoder.json"), "w") as f:
        json.dump({"cases": cases}, f)
    print("wrote external_encoder.json")


def pin_tower(ref_dir):
    import numpy as np
    sys.path.insert(0, os.path.join(ref_dir, "src", "chessrl"))
    model_mod = importlib.import_module("model")
    m = model_mod.ChessModel(compile_model=False)           # the reference's own topology (10 x 256)
    rng = np.random.default_rng(0)
    x = (rng.random((4, 8, 8, 127)) < 0.12).astype(np.float32)
    pol, val = m.predict(x)
    names, arrays = [], {}
    for layer in m.model.layers:
        for w in layer.weights:
            names.append(w.name)
            arrays["w%04d" % (len(names) - 1)] = w.numpy()
    np.savez_compressed(os.path.join(OUT, "external_tower.npz"), x=x, policy=pol, value=val,
                        weight_names=np.array(names), **arrays)
    print("wrote external_tower.npz (%d weight tensors)" % len(names))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    done = []
    for name, fn in (("rules", pin_rules), ("encoder", pin_encoder), ("tower", pin_tower)):
        try:
            fn(a.reference)
            done.append(name)
        except ImportError as e:
            print("skipped %s: %s" % (name, e))
    print("pinned:", done or "nothing (python-chess / tensorflow do not import here)")


if __name__ == "__main__":
    main()
