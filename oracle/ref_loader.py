"""oracle/ref_loader.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Imports the reference's own ``mctree.py`` from /root/reference (present in the
authoring container only, never on the GPU box) so that the oracle can be
validated against it and golden vectors generated.  ``mctree.py`` imports
``from game import Game`` (mctree.py:3), which needs python-chess; a stub
``game`` module whose ``Game`` is the C-oracle-backed duck type is injected
instead.  Nothing is copied: the reference file is
executed where it lies.
"""
import importlib
import os
import sys
import types

REF_DIR = "/root/reference/src/chessrl"


def available():
    return os.path.exists(os.path.join(REF_DIR, "mctree.py"))


def load_mctree():
    from .chess_oracle import OracleGame
    stub = types.ModuleType("game")
    stub.Game = OracleGame
    saved = {k: sys.modules.get(k) for k in ("game", "player", "mctree")}
    sys.modules["game"] = stub
    sys.path.insert(0, REF_DIR)
    try:
        for k in ("player", "mctree"):
            sys.modules.pop(k, None)
        mod = importlib.import_module("mctree")
    finally:
        sys.path.remove(REF_DIR)
        for k, v in saved.items():
            if k == "mctree":
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("mctree", None)
    return mod


def load_uci_labels():
    """exec the pure function netencoder.get_uci_labels (netencoder.py:94-134)."""
    import ast
    src = open(os.path.join(REF_DIR, "netencoder.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_uci_labels"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "netencoder.py", "exec"), ns)
    return ns["get_uci_labels"]()


def _extract(path, name, cls=None):
    """The AST node of function `name` (a method of class `cls` if given) of a reference file."""
    import ast
    tree = ast.parse(open(os.path.join(REF_DIR, path)).read())
    body = tree.body
    if cls is not None:
        body = [n for n in body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    return [n for n in body if isinstance(n, ast.FunctionDef) and n.name == name][0]


def load_play_game(mct):
    """(play_game, best_move): the reference's own ``selfplay.play_game`` (selfplay.py:59-84) and
    ``AgentDistributed.best_move`` (agentdistributed.py:39-68), executed where they lie.  Their modules
    import TensorFlow / python-chess at the top, so the two functions are taken out of the parsed
    files and run in a namespace holding what they reference: the reference's ``mctree`` (load_mctree),
    ``Game`` = the C-oracle duck type, a silent ``Logger``, ``random``, ``timer`` and ``np``."""
    import ast
    import random
    from timeit import default_timer as timer
    import numpy as np
    from .chess_oracle import OracleGame

    class Logger(object):
        @staticmethod
        def get_instance():
            return Logger()

        def debug(self, *a, **k):
            pass

    ns = {"Game": OracleGame, "Logger": Logger, "random": random, "timer": timer, "np": np, "mctree": mct}
    for node, fname in ((_extract("selfplay.py", "play_game"), "selfplay.py"),
                        (_extract("agentdistributed.py", "best_move", cls="AgentDistributed"), "agentdistributed.py")):
        exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF_DIR, fname), "exec"), ns)
    return ns["play_game"], ns["best_move"]
