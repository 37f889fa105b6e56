"""``PredictWorker`` -- what is left of the reference's prediction server.

The reference serves tower evaluations to the search threads over a localhost TCP socket
(/root/reference/src/chessrl/predict_worker.py:12-128: listener thread, opportunistic batching
of at most ``threads`` requests, float16 cast, ``model.predict``).  Here the tower runs in the
search's own process on the same HIP stream and the batch is every game on the GPU, so there is
nothing to serve: this class only keeps the ``start`` / ``stop`` / ``reload_model`` surface that
``selfplay.main`` drives (selfplay.py:137,143,154), as a holder of the model.
"""
from .model import ChessModel


class PredictWorker(object):
    def __init__(self, model_path=None, endpoint=("localhost", 9999), **model_kwargs):
        self.model_kwargs = model_kwargs
        self.model = ChessModel(weights=model_path, **model_kwargs)
        self.address = endpoint            # accepted and ignored: no socket exists
        self.running = False

    def start(self):
        self.running = True

    def stop(self):
        self.running = False

    def reload_model(self, model_path):
        """Load new weights (predict_worker.py:63-70); the reference never calls it in selfplay.py."""
        self.model = ChessModel(weights=model_path, **self.model_kwargs)
