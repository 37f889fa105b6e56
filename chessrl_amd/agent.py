"""``Agent`` / ``AgentDistributed`` -- host mirror of the reference's players.

Same surface as /root/reference/src/chessrl/agentdistributed.py:28-111 (and
agent.py:20-62): ``best_move``, ``predict_policy``, ``predict_outcome``, ``predict``,
``get_copy``, ``connect``, ``disconnect``; attributes ``color``, ``move_encodings``,
``uci_dict``.  The reference ships a request over TCP to a PredictWorker
(predict_worker.py:72-111); here the tower lives in the same process on the same GPU,
so ``connect``/``disconnect`` are no-ops and ``endpoint`` is accepted and ignored.

``best_move(real_game=False)`` builds a fresh ``SelfPlayTree`` per move exactly like
agentdistributed.py:61-66.  (The reference's local ``Agent`` calls the stub ``Tree`` and
returns None, agent.py:46-47; here both names run the working search.)
"""
import numpy as np

from . import netencoder
from .player import Player


class Agent(Player):
    """``model``: a ``ChessModel`` (or any callable planes -> (policy, value) on the GPU)."""

    def __init__(self, color, weights=None, endpoint=None, num_threads=6, model=None,
                 blocks=10, filters=256, numpy_promotion="auto"):
        super().__init__(color)
        if model is None:
            from .model import ChessModel
            model = ChessModel(compile_model=True, weights=weights, blocks=blocks, filters=filters)
        self.model = model
        self.move_encodings = netencoder.get_uci_labels()
        self.uci_dict = {u: i for i, u in enumerate(self.move_encodings)}
        self.address = endpoint
        self.num_threads = num_threads
        self.numpy_promotion = numpy_promotion
        self._engines = {}

    # ---- tower requests (agentdistributed.py:70-99) --------------------------------------
    def _eval(self, game):
        import torch
        from .game import arena
        a = arena()
        planes = a.planes()
        ctx = a.one(game._slot)
        ctx.set_stream(torch.cuda.current_stream(planes.device).cuda_stream)
        ctx.encode(planes.data_ptr())
        pol, val = self.model(planes)
        return pol[0].float().cpu().numpy(), float(val[0])

    def predict(self, game):
        return self._eval(game)

    def predict_outcome(self, game):
        return self._eval(game)[1]

    def predict_policy(self, game, mask_legal_moves=True):
        policy = self._eval(game)[0]
        if mask_legal_moves:
            policy = [policy[self.uci_dict[x]] for x in game.get_legal_moves()]
        return policy

    # ---- moves -------------------------------------------------------------------------------
    def best_move(self, game, real_game=False, max_iters=900, ai_move=True, verbose=False):
        best_move = "00000"
        if real_game:
            policy = self.predict_policy(game)
            best_move = game.get_legal_moves()[int(np.argmax(policy))]
        elif game.get_result() is None:
            from . import mctree
            tree = mctree.SelfPlayTree(game, threads=self.num_threads)
            best_move = tree.search_move(self, max_iters=max_iters, verbose=verbose, ai_move=ai_move)
        return best_move

    def engine_for(self, max_iters):
        """One single-game LockstepEngine per simulation budget, reused across moves."""
        from .engine import LockstepEngine
        if max_iters not in self._engines:
            self._engines[max_iters] = LockstepEngine(
                self.model, n_games=1, max_sims=max_iters, numpy_promotion=self.numpy_promotion)
        return self._engines[max_iters]

    def get_copy(self):
        return self

    def connect(self):
        pass

    def disconnect(self):
        pass

    def train(self, dataset, epochs=1, logdir=None, batch_size=1, validation_split=0):
        """Fit the model on recorded games (agent.py:64-89).  The last
        ``validation_split`` fraction of the games is held out for validation; training batches
        are flipped at random with probability 0.1.  Nothing happens on an empty dataset."""
        from .dataset import DatasetGame
        n = len(dataset)
        if n == 0:
            return None
        held_out = int(validation_split * n) if validation_split > 0 else 0
        val_gen = None
        if validation_split > 0:
            val_gen = netencoder.DataGameSequence(DatasetGame(dataset[n - held_out:]), batch_size=batch_size)
            dataset = DatasetGame(dataset[:n - held_out])
        fit_gen = netencoder.DataGameSequence(dataset, batch_size=batch_size, random_flips=.1)
        return self.model.train_generator(fit_gen, epochs=epochs, logdir=logdir, val_gen=val_gen)

    def save(self, path):
        self.model.save_weights(path)

    def load(self, path):
        self.model.load_weights(path)


AgentDistributed = Agent
