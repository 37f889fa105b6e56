"""oracle/mcts_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (pure Python + numpy scalars) of the reference's self-play
search, sequential ``threads=1`` semantics -- the only deterministic mode of
/root/reference/src/chessrl/mctree.py -- plus the agent
and game-loop glue around it:

  search          <- SelfPlayTree.search_move / explore_tree   mctree.py:159-214
  _select_expand  <- select + expand + Node.__init__           mctree.py:216-257, 28-37
  _puct           <- Node.get_value / get_best_child           mctree.py:71-95
  _backup         <- simulate + backprop                       mctree.py:259-296
  compute_policy  <- SelfPlayTree.compute_policy               mctree.py:305-322
  OracleAgent     <- AgentDistributed.best_move/predict_*      agentdistributed.py:39-83
  play_game       <- selfplay.play_game                        selfplay.py:59-84

PARITY STATUS: pinned.  tests/test_oracle_mcts.py runs the reference's own
mctree.py (imported from /root/reference with a stub ``game`` module) against
this restatement on the same games and nets, and tests/golden/mcts_*.json
holds its outputs (made by oracle/make_golden.py) for boxes without the
reference.

Two float modes exist because the reference pins numpy==1.17.2
(requirements.txt:6) while this image has numpy 2.x:
  "nep50"  -- ``10 * np.float32(prior)`` stays float32 (numpy >= 2; what the
              reference's mctree.py does when run in this container);
  "legacy" -- the same product is float64 (numpy 1.x scalar promotion).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np

from . import encoder_oracle
from .chess_oracle import OracleGame, NULL_MOVE

RESULT_NONE = None


class OracleAgent(object):
    """AgentDistributed stand-in: encoder oracle + a net callable, batch of 1.

    ``net(planes[B,8,8,127] torch tensor) -> (policy[B,1968] f32, value[B] f32)``.
    """

    def __init__(self, net, color=True, widen_priors=False):
        self.net = net
        self.color = color
        self.widen_priors = widen_priors   # emulate numpy-1.x promotion for mctree.py
        self.move_encodings = encoder_oracle.get_uci_labels()
        self.uci_dict = {u: i for i, u in enumerate(self.move_encodings)}
        self.n_evals = 0

    def _eval(self, game):
        import torch
        planes = torch.from_numpy(encoder_oracle.get_game_state(game)[None])
        pol, val = self.net(planes)
        self.n_evals += 1
        return pol[0].cpu().numpy().astype(np.float32), float(val[0])

    def predict(self, game):
        return self._eval(game)

    def predict_outcome(self, game):
        return self._eval(game)[1]

    def predict_policy(self, game, mask_legal_moves=True):
        policy = self._eval(game)[0]
        if mask_legal_moves:
            policy = [policy[self.uci_dict[x]] for x in game.get_legal_moves()]
            if self.widen_priors:
                policy = [np.float64(p) for p in policy]
        return policy

    def best_move(self, game, real_game=False, max_iters=900, ai_move=True, verbose=False,
                  noise=True, mode="nep50", rng=None):
        best = NULL_MOVE
        if real_game:
            policy = self.predict_policy(game)
            best = game.get_legal_moves()[int(np.argmax(policy))]
        elif game.get_result() is None:
            best = search(game, self, max_iters, noise=noise, ai_move=ai_move, mode=mode,
                          rng=rng).moves
        return best

    def get_copy(self):
        return self

    def connect(self):
        pass

    def disconnect(self):
        pass


class SearchResult(object):
    def __init__(self):
        self.moves = None            # (bm, am) as search_move(ai_move=True) returns them
        self.visits = None           # root children visits, CHILDREN order
        self.values = None           # root children value sums (f64), children order
        self.priors = None           # root children priors (f32), children order
        self.child_moves = None      # our move of each root child, children order
        self.child_replies = None    # stored opponent reply (or NULL_MOVE), children order
        self.policy = None
        self.chosen = None
        self.root_visits = None
        self.n_nodes = None
        self.max_depth = 0


class _N(object):
    __slots__ = ("state", "parent", "kids", "todo", "value", "visits", "prior", "result",
                 "move", "reply")


def _new_node(state, parent):
    n = _N()
    n.state, n.parent, n.kids = state, parent, []
    n.todo = state.get_legal_moves()          # Node.__init__ mctree.py:31
    n.value, n.visits, n.prior = 0, 0, 1
    n.result = state.get_result()             # is_terminal_state mctree.py:47-49 (cached)
    n.move = n.reply = NULL_MOVE
    return n


def _puct(c, mode):
    """mctree.py:71-87 with the numpy scalar types spelled out."""
    q = c.value / (1 + c.visits)                                      # python float
    sumv = np.sum([g.visits for g in c.kids])                         # int64 (0.0 if empty)
    if mode == "legacy" and isinstance(c.prior, np.float32):
        cp = np.float64(10) * np.float64(c.prior)                     # numpy 1.x: f64 product
    else:
        cp = 10 * c.prior                                             # numpy 2.x: stays f32
    return q + cp * (np.sqrt(sumv) / (1 + c.visits))


def search(game, agent, max_iters, noise=False, ai_move=True, mode="nep50", rng=None):
    """One SelfPlayTree(game).search_move(agent, max_iters, noise=..., ai_move=True)."""
    res = SearchResult()
    root = _new_node(game.get_copy(), None)
    root.visits = 1                                                    # mctree.py:111
    n_nodes = 1
    for _ in range(max_iters):
        node, depth = root, 0
        while node.result is None:                                     # select mctree.py:218
            if node.todo:                                              # not fully expanded
                st = node.state.get_copy()
                mv = node.todo.pop()                                   # LAST legal move first
                st.move(mv)
                reply = NULL_MOVE
                if st.get_result() is None:                            # opponent's greedy reply
                    reply = agent.best_move(st, real_game=True)
                    st.move(reply)
                child = _new_node(st, node)
                child.move, child.reply = mv, reply
                node.kids.append(child)
                n_nodes += 1
                if not node.todo:                                      # _update_prior mctree.py:298
                    pri = agent.predict_policy(node.state, mask_legal_moves=True)
                    for p, k in zip(pri, reversed(node.kids)):
                        k.prior = p
                node = child
                depth += 1
                break
            vals = [_puct(c, mode) for c in node.kids]
            node = node.kids[int(np.argmax(vals))]                     # first max
            depth += 1
        res.max_depth = max(res.max_depth, depth)
        v = node.result                                                # simulate mctree.py:268
        if v is None:
            v = agent.predict_outcome(node.state)
        while node is not None:                                        # backprop: same v, no flip
            node.visits += 1
            node.value += v
            node = node.parent
    res.visits = [c.visits for c in root.kids]
    res.values = [float(c.value) for c in root.kids]
    res.priors = [float(c.prior) for c in root.kids]
    res.child_moves = [c.move for c in root.kids]
    res.child_replies = [c.reply for c in root.kids]
    res.root_visits = root.visits
    res.n_nodes = n_nodes
    res.policy = compute_policy(res.visits, root.visits, len(game), noise=noise, rng=rng)
    res.chosen = int(np.argmax(res.policy))
    ch = root.kids[res.chosen]
    stack = ch.state.board.move_stack
    bm = am = NULL_MOVE
    if len(stack) >= 2:                                                # mctree.py:185-194
        bm, am = str(stack[-2]), str(stack[-1])
    res.moves = (bm, am) if ai_move else bm
    return res


def compute_policy(visits, root_visits, nb_moves, noise=True, rng=None):
    """mctree.py:305-322 (rng=None -> global np.random like the reference)."""
    tau = 1
    if nb_moves >= 30:
        tau = nb_moves / (1 + np.power(nb_moves, 1.3))
    policy = np.array([np.power(v, 1 / tau) for v in visits]) / np.power(root_visits, 1 / tau)
    if noise:
        eps = 0.25
        d = (rng or np.random).dirichlet([0.03] * len(visits))
        policy = (1 - eps) * policy + d
    return policy


def play_game(agent, max_iters=900, noise=True, mode="nep50", player_color=None,
              pyrandom=None, max_moves=None, rng=None):
    """selfplay.play_game (selfplay.py:59-84) on OracleGame; returns the Game."""
    import random
    if player_color is None:
        player_color = True if (pyrandom or random).random() >= 0.5 else False
    gam = OracleGame(player_color=player_color)
    agent.color = player_color
    if player_color is False:
        gam.move(agent.best_move(gam, real_game=True))
    n = 0
    while gam.get_result() is None:
        bm, am = agent.best_move(gam, real_game=False, ai_move=True, max_iters=max_iters,
                                 noise=noise, mode=mode, rng=rng)
        gam.move(bm)
        gam.move(am)
        n += 1
        if max_moves is not None and n >= max_moves:
            break
    return gam
