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
