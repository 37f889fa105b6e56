"""``SelfPlayTree`` -- host mirror of the reference's MCTS class.

Same surface as /root/reference/src/chessrl/mctree.py:148-198: construct from a ``Game``,
call ``search_move(agent, max_iters, verbose, noise, ai_move)``, then read
``tree.root.visits`` and ``tree.root.children[i].visits / .value / .prior``.  The tree
itself lives in HBM (flat node/edge arrays, one wavefront per game); select / expand /
backup are the HIP kernels behind crl_sim_* and the final ``compute_policy``
(mctree.py:305-322) stays on the host in numpy so that the Dirichlet noise comes from the
same ``np.random`` stream as the reference's.

``threads`` is accepted for signature compatibility; simulations run with the
reference's sequential (threads=1) semantics, its only deterministic mode.
"""
import numpy as np

from .engine import compute_policy
from .game import Game, move_to_uci
from . import _lib


class Node(object):
    """Read-only view of one root child (``state`` is not materialised)."""

    def __init__(self, visits, value, prior, move, reply):
        self.visits, self.value, self.prior = int(visits), float(value), np.float32(prior)
        self.move, self.reply = move, reply
        self.vloss = 0
        self.children = []


class _Root(object):
    def __init__(self, visits, children):
        self.visits, self.children = int(visits), children
        self.parent = None


class Tree(object):
    def __init__(self, root):
        if not isinstance(root, Game):
            raise TypeError("root must be a chessrl_amd Game")
        self._game = root
        self.root = _Root(1, [])


class SelfPlayTree(Tree):

    def __init__(self, root, threads=6):
        super().__init__(root)
        self.num_threads = threads

    def search_move(self, agent, max_iters=200, verbose=False, noise=True, ai_move=False):
        eng = agent.engine_for(max_iters)
        ids = self._game.move_ids()
        eng.load_moves([list(ids)])
        eng.search(max_iters)
        rc = eng.root_children()
        n = int(rc["nchild"][0])
        kids = [Node(rc["visits"][0, k], rc["values"][0, k], rc["priors"][0, k],
                     move_to_uci(rc["moves"][0, k]),
                     None if rc["replies"][0, k] == _lib.NO_MOVE else move_to_uci(rc["replies"][0, k]))
                for k in range(n)]
        self.root = _Root(rc["root_visits"][0], kids)
        policy = compute_policy([c.visits for c in kids], self.root.visits, len(ids), noise=noise)
        best = kids[int(np.argmax(policy))]
        if best.reply is not None:
            b_mov, agent_last_mov = best.move, best.reply
        else:
            # the game ended on our move: the reference returns move_stack[-2:] of the child,
            # i.e. (previous ply, our move) (mctree.py:185-188), or two NULL moves on IndexError
            if len(ids) >= 1:
                b_mov, agent_last_mov = move_to_uci(ids[-1]), best.move
            else:
                b_mov = agent_last_mov = Game.NULL_MOVE
        return (b_mov, agent_last_mov) if ai_move else b_mov
