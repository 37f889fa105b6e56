"""``SelfPlayTree`` -- host mirror of the reference's MCTS class.

Same surface as /root/reference/src/chessrl/mctree.py:148-198: construct from a ``Game``,
call ``search_move(agent, max_iters, verbose, noise, ai_move)``, then read
``tree.root.visits`` and ``tree.root.children[i].visits / .value / .prior / .state``.  The
tree itself lives in HBM (flat node/edge arrays, one wavefront per game); select / expand /
backup are the HIP kernels behind crl_sim_* and the final ``compute_policy``
(mctree.py:305-322) stays on the host in numpy so that the Dirichlet noise comes from the
same ``np.random`` stream as the reference's.

The root is the caller's game as it stands -- whatever position it was set up from (standard
start, FEN, board row) and whatever moves were pushed since: the search engine's slot is a
device-side deep copy of the Game's arena slot (``crl_copy_game_from``), the counterpart of
``Node(root.get_copy())`` in mctree.py:105-109.

``threads`` is accepted for signature compatibility; simulations run with the
reference's sequential (threads=1) semantics, its only deterministic mode.
"""
import numpy as np

from .engine import compute_policy
from .game import Game, arena, move_to_uci
from . import _lib


class Node(object):
    """Read-only view of one root child.  ``state`` (the child's Game: root + our move + the
    stored reply) is built on first access from the tree's SNAPSHOT of the root -- the copy taken
    when the search ran, mctree.py:105-109 ``Node(root.get_copy())`` -- so it does not depend on
    what the caller did to its own game afterwards; the device tree keeps only boards."""

    def __init__(self, root_snapshot, visits, value, prior, move, reply):
        self.visits, self.value, self.prior = int(visits), float(value), np.float32(prior)
        self.move, self.reply = move, reply
        self.vloss = 0
        self.children = []
        self._root_snapshot, self._state = root_snapshot, None

    @property
    def state(self):
        if self._state is None:
            g = self._root_snapshot.get_copy()
            for mv in (self.move, self.reply):
                if mv is not None and not g.move(mv):
                    raise RuntimeError("tree child %s%s does not replay on the root snapshot"
                                       % (self.move, "+" + self.reply if self.reply else ""))
            self._state = g
        return self._state


class _Root(object):
    def __init__(self, game, visits, children):
        self.state, self.visits, self.children = game, int(visits), children
        self.parent = None


class Tree(object):
    def __init__(self, root):
        if not isinstance(root, Game):
            raise TypeError("root must be a chessrl_amd Game")
        self._game = root
        self.root = _Root(root.get_copy(), 1, [])        # mctree.py:105-109: Node(root.get_copy())


class SelfPlayTree(Tree):

    def __init__(self, root, threads=6):
        super().__init__(root)
        self.num_threads = threads

    def search_move(self, agent, max_iters=200, verbose=False, noise=True, ai_move=False):
        # the tree searches its own snapshot of the caller's game (taken at construction, as the
        # reference's Node(root.get_copy())); one arena slot, returned when the tree is collected
        game = self.root.state
        eng = agent.engine_for(max_iters)
        eng.ctx.copy_game_from(0, arena().ctx, game._slot)
        eng.search(max_iters)
        rc = eng.root_children()
        n = int(rc["nchild"][0])
        if n == 0:
            # a finished root has no children; the reference's np.argmax([]) raises here too
            raise ValueError("search_move on a finished game (attempt to get argmax of an empty sequence)")
        kids = []
        for k in range(n):
            reply = rc["replies"][0, k]
            kids.append(Node(game, rc["visits"][0, k], rc["values"][0, k], rc["priors"][0, k],
                             move_to_uci(rc["moves"][0, k]),
                             None if reply == _lib.NO_MOVE else move_to_uci(reply)))
        self.root = _Root(game, rc["root_visits"][0], kids)
        stack = game.move_ids()
        policy = compute_policy([c.visits for c in kids], self.root.visits, len(stack), noise=noise)
        best = kids[int(np.argmax(policy))]
        # mctree.py:185-194 reads the last two entries of the chosen child's move stack
        if best.reply is not None:
            pair = (best.move, best.reply)
        elif len(stack) >= 1:                    # game over after our move: (previous ply, our move)
            pair = (move_to_uci(stack[-1]), best.move)
        else:                                    # one-entry stack: the IndexError branch
            pair = (Game.NULL_MOVE, Game.NULL_MOVE)
        return pair if ai_move else pair[0]
