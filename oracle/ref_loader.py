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


def load_dataset():
    """The reference's own ``dataset.py`` (it imports only ``game`` and ``json``), imported where it
    lies with the same stub ``game`` module as mctree (Game = the C-oracle duck type)."""
    from .chess_oracle import OracleGame
    stub = types.ModuleType("game")
    stub.Game = OracleGame
    saved = {k: sys.modules.get(k) for k in ("game", "dataset")}
    sys.modules["game"] = stub
    sys.modules.pop("dataset", None)
    sys.path.insert(0, REF_DIR)
    try:
        mod = importlib.import_module("dataset")
    finally:
        sys.path.remove(REF_DIR)
        sys.modules.pop("dataset", None)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
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
    nodes = [(_extract("selfplay.py", "play_game"), "selfplay.py")]
    for meth in AGENT_METHODS:
        nodes.append((_extract("agentdistributed.py", meth, cls="AgentDistributed"), "agentdistributed.py"))
    for node, fname in nodes:
        exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF_DIR, fname), "exec"), ns)
    return ns["play_game"], ns["best_move"]


# AgentDistributed methods executed from the reference (agentdistributed.py:39-90); compiled outside
# their class, so ``self.__send_game`` is looked up under that literal name
AGENT_METHODS = ("best_move", "predict_outcome", "predict_policy", "predict")


def make_reference_agent(mct, net, sims):
    """An agent whose best_move / predict_policy / predict_outcome / predict ARE the reference's
    (see load_play_game), whose ``uci_dict`` is built from the reference's get_uci_labels and whose
    ``__send_game`` -- the TCP round trip to the prediction worker in the reference -- encodes the game
    with the reference's own get_game_state (load_encoder) and evaluates ``net`` on it.  ``best_move``
    runs with ``sims`` simulations whatever max_iters the caller passes (play_game says 900)."""
    import ast
    import numpy as np
    import torch
    ns = {"np": np, "mctree": mct}
    for meth in AGENT_METHODS:
        node = _extract("agentdistributed.py", meth, cls="AgentDistributed")
        exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF_DIR, "agentdistributed.py"), "exec"), ns)
    encode = load_encoder()
    labels = load_uci_labels()
    ref_best_move = ns["best_move"]

    class ReferenceAgent(object):
        num_threads = 1
        predict_outcome, predict_policy, predict = ns["predict_outcome"], ns["predict_policy"], ns["predict"]

        def __init__(self, color=True):
            self.color = color
            self.move_encodings = labels
            self.uci_dict = {u: i for i, u in enumerate(labels)}
            self.n_evals = 0

        def best_move(self, game, real_game=False, max_iters=900, ai_move=True, verbose=False):
            return ref_best_move(self, game, real_game=real_game, max_iters=sims, ai_move=ai_move, verbose=verbose)

        def _send(self, game):
            pol, val = net(torch.from_numpy(np.asarray(encode(game))[None]))
            self.n_evals += 1
            return pol[0].cpu().numpy().astype(np.float32), float(val[0])

        def get_copy(self):
            return self

        def connect(self):
            pass

        def disconnect(self):
            pass

    setattr(ReferenceAgent, "__send_game", ReferenceAgent._send)
    return ReferenceAgent()


class _SquareSet(object):
    """What netencoder.py:25-26 touches of a python-chess SquareSet: ``mirror()`` (the documented
    'vertically mirrored copy': rank r <-> rank 9-r) and ``tolist()`` (64 bools, squares a1..h8)."""

    def __init__(self, bb):
        self.bb = bb & 0xFFFFFFFFFFFFFFFF

    def mirror(self):
        return _SquareSet(int.from_bytes(self.bb.to_bytes(8, "little"), "big"))

    def tolist(self):
        return [bool((self.bb >> sq) & 1) for sq in range(64)]


class _BoardAdapter(object):
    """The four python-chess ``Board`` members the reference's encoder uses (netencoder.py:25,57,62;
    ``pieces(piece_type, color)`` with PAWN..KING = 1..6, ``copy()``, ``pop()`` raising IndexError on
    an empty move stack) over the C oracle's position history.  This adapter is the part of the
    encoder pin that is NOT the reference's code."""

    def __init__(self, game, back=0):
        self._g, self._back = game, back

    def pieces(self, piece_type, color):
        b = self._g.board_at(self._back)
        occ = 0
        for t in range(6):
            occ |= int(b.bb[t])
        own = int(b.white) if color else (occ & ~int(b.white))
        return _SquareSet(int(b.bb[piece_type - 1]) & own)

    def copy(self):
        return _BoardAdapter(self._g, self._back)

    def pop(self):
        if self._back + 1 > len(self._g):
            raise IndexError("pop from empty list")
        self._back += 1


class _GameAdapter(object):
    def __init__(self, game):
        self.board, self.turn = _BoardAdapter(game), game.turn


def load_encoder():
    """The reference's own ``get_game_state`` and its helpers (netencoder.py:13-91), taken out of
    the parsed file (its imports need python-chess and TensorFlow) and executed in a namespace with
    ``np`` and a ``chess`` that only has PIECE_TYPES = 1..6; call it on an OracleGame."""
    import ast
    import numpy as np
    ns = {"np": np, "chess": types.SimpleNamespace(PIECE_TYPES=range(1, 7))}
    nodes = [_extract("netencoder.py", n) for n in
             ("_get_pieces_one_hot", "_get_current_game_state", "_get_game_history", "get_game_state")]
    exec(compile(ast.Module(body=nodes, type_ignores=[]), os.path.join(REF_DIR, "netencoder.py"), "exec"), ns)
    fn = ns["get_game_state"]
    return lambda game, flipped=False: fn(_GameAdapter(game), flipped=flipped)


def load_data_sequence(ds_mod):
    """The reference's own training-batch generator ``DataGameSequence`` (netencoder.py:137-181),
    taken out of the parsed file and executed with the reference's get_game_state (over the adapter
    above), get_uci_labels, ``Sequence = object`` and Keras' documented ``to_categorical`` (a float32
    one-hot vector) in its namespace.  Games are the C-oracle duck type, wrapped for the encoder."""
    import ast
    import numpy as np
    encode = load_encoder()

    def to_categorical(y, num_classes=None):
        v = np.zeros(num_classes, dtype=np.float32)
        v[int(y)] = 1.0
        return v

    tree = ast.parse(open(os.path.join(REF_DIR, "netencoder.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DataGameSequence"][0]
    ns = {"np": np, "Sequence": object, "to_categorical": to_categorical, "get_uci_labels": load_uci_labels,
          "get_game_state": lambda game, flipped=False: encode(game, flipped=flipped)}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), os.path.join(REF_DIR, "netencoder.py"), "exec"), ns)
    return ns["DataGameSequence"]


def record_model_graph():
    """Execute the reference's own ``ChessModel.__init__`` and ``__res_block`` (model.py:17-72,111-122)
    over RECORDING stand-ins of the Keras constructors they call (Input, Conv2D, BatchNormalization,
    Activation, Flatten, Dense, Add, Model, Adam): no arithmetic, only which layer is built with which
    arguments on which inputs, and what ``compile`` is given.  Returns {"nodes": [...], "outputs": [...],
    "compile": {...}} -- the topology and hyper-parameters as the reference code states them (what
    Keras then does with 'l2', BatchNormalization() defaults etc. is Keras documentation, not pinned)."""
    import ast
    nodes = []

    class T(object):                                    # a symbolic tensor = the id of the node that made it
        def __init__(self, i):
            self.id = i

    def layer(op):
        def ctor(*args, **kw):
            cfg = dict(kw)
            if args:
                cfg["args"] = [list(a) if isinstance(a, tuple) else a for a in args]

            def call(x):
                ins = [t.id for t in x] if isinstance(x, (list, tuple)) else [x.id]
                nodes.append({"op": op, "config": cfg, "inputs": ins})
                return T(len(nodes) - 1)
            return call
        return ctor

    def Input(shape):
        nodes.append({"op": "Input", "config": {"shape": list(shape)}, "inputs": []})
        return T(len(nodes) - 1)

    rec = {}

    class Model(object):
        def __init__(self, inp, outs):
            rec["inputs"], rec["outputs"] = [inp.id], [o.id for o in outs]

        def load_weights(self, path):
            rec["load_weights"] = path

        def compile(self, optimizer, loss=None, metrics=None):
            rec["compile"] = {"optimizer": optimizer, "loss": loss, "metrics": metrics}

    ns = {"Input": Input, "Model": Model, "Adam": lambda **kw: {"class": "Adam", **kw}}
    for op in ("Conv2D", "BatchNormalization", "Activation", "Flatten", "Dense", "Add"):
        ns[op] = layer(op)
    init = _extract("model.py", "__init__", cls="ChessModel")
    blk = _extract("model.py", "__res_block", cls="ChessModel")
    exec(compile(ast.Module(body=[init, blk], type_ignores=[]), os.path.join(REF_DIR, "model.py"), "exec"), ns)

    class Self(object):
        pass
    me = Self()
    setattr(Self, "__res_block", ns["__res_block"])      # compiled outside the class: no name mangling
    ns["__init__"](me, compile_model=True, weights=None)
    return {"nodes": nodes, "inputs": rec["inputs"], "outputs": rec["outputs"], "compile": rec["compile"]}


def load_game_agent():
    """(GameAgent class, Agent marker class): the reference's own ``GameAgent`` (gameagent.py:7-50),
    its class statement taken out of the parsed file and executed with ``Game`` = the C-oracle duck
    type whose ``board`` also has ``push`` (python-chess ``Board.push`` of a legal move), ``chess.Move.
    from_uci`` = the identity on UCI strings, and ``Agent`` = a marker class the test agent derives from."""
    import ast
    from .chess_oracle import OracleGame

    class _PushBoard(object):
        def __init__(self, game):
            self._g, self._view = game, game.board

        @property
        def move_stack(self):
            return self._view.move_stack

        @property
        def turn(self):
            return self._view.turn

        def push(self, uci):
            assert OracleGame.move(self._g, uci), uci

        def copy(self):
            return self._g

    class Game(OracleGame):
        def __init__(self, board=None, player_color=True, date=None):
            if isinstance(board, OracleGame):            # gameagent.py:46: GameAgent(board=self.board.copy(), ...)
                src = OracleGame.get_copy(board)
                OracleGame.__init__(self, player_color=player_color, date=date, _handle=src._h)
                src._h = None
            else:
                OracleGame.__init__(self, board=board, player_color=player_color, date=date)
            self.board = _PushBoard(self)

    class Agent(object):
        pass

    tree = ast.parse(open(os.path.join(REF_DIR, "gameagent.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GameAgent"][0]
    ns = {"Game": Game, "Agent": Agent, "chess": types.SimpleNamespace(Move=types.SimpleNamespace(from_uci=lambda u: u))}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), os.path.join(REF_DIR, "gameagent.py"), "exec"), ns)
    return ns["GameAgent"], Agent
