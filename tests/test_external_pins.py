"""Consumers of the fixtures oracle/pin_external.py writes when python-chess 0.28.3 / TensorFlow are
available (they are not in this image: every test here then skips).  CPU: the C oracle against the
reference-on-python-chess outputs; GPU: the HIP rules / encoder / tower against the same."""
import json
import os

import numpy as np
import pytest

from oracle.chess_oracle import OracleGame, board_from_fen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip("%s not generated yet (python -m oracle.pin_external on a machine with python-chess / "
                    "tensorflow)" % name)
    return json.load(open(path))


def test_oracle_rules_match_python_chess_fixture():
    ext = _load("external_rules.json")
    for p in ext["positions"]:
        g = OracleGame(board=board_from_fen(p["fen"]))
        assert g.get_legal_moves() == p["legal"], p["fen"]             # ORDER, not just the set
        assert g.get_result() == p["result"], p["fen"]
    for game in ext["games"]:
        g = OracleGame()
        for ply in game["plies"]:
            assert g.get_legal_moves() == ply["legal"], (game["seed"], len(g))
            assert g.move(ply["move"])
            assert g.get_result() == ply["result_after"]
            st, b = ply["state_after"], g.board_at(0)
            assert g.get_fen() == st["board_fen"] and bool(b.state & 1) == st["turn"]
            assert (b.state >> 12) & 255 == min(st["clock"], 255)
            assert ((b.state >> 20) & 1) == int(st["has_legal_ep"])
    g = OracleGame()
    for step in ext["repetition"]:
        g.move(step["move"])
        assert g.get_result() == step["result_after"]


@pytest.mark.gpu
def test_hip_rules_match_python_chess_fixture():
    from chessrl_amd.game import Game
    ext = _load("external_rules.json")
    for p in ext["positions"]:
        g = Game(board=p["fen"])
        assert g.get_legal_moves() == p["legal"], p["fen"]
        assert g.get_result() == p["result"], p["fen"]
        g.free()
    for game in ext["games"]:
        g = Game()
        for ply in game["plies"]:
            assert g.get_legal_moves() == ply["legal"], (game["seed"], len(g))
            assert g.move(ply["move"]) and g.get_result() == ply["result_after"]
        g.free()


def test_oracle_encoder_matches_reference_fixture():
    from oracle import encoder_oracle
    ext = _load("external_encoder.json")
    for c in ext["cases"]:
        g = OracleGame()
        for u in c["moves"]:
            assert g.move(u)
        planes = encoder_oracle.get_game_state(g, flipped=c["flipped"])
        assert list(planes.shape) == c["shape"] and c["all_binary"]
        assert np.flatnonzero(planes.reshape(-1) != 0).tolist() == c["ones"]
