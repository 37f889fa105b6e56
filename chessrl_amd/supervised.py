"""Supervised training on recorded games -- host mirror of the reference's ``supervised.py``.

Same CLI as /root/reference/src/chessrl/supervised.py:65-95 (``modeldir datadir --epochs --bs
--debug``) and the same ``train`` (supervised.py:37-62): load a ``DatasetGame`` JSON, load or
create the newest model of ``modeldir``, ``Agent.train`` with ``validation_split=0.25``, save.
Weights are ``.npz`` (the reference's ``.h5`` needs h5py); ``--blocks/--filters`` size a fresh
model (the reference hard-codes 10 x 256).
"""
import argparse
import logging
import os

from .agent import Agent
from .dataset import DatasetGame
from .selfplay import get_model_path

log = logging.getLogger("chessrl_amd.supervised")


def train(model_dir, dataset_path, epochs=1, batch_size=8, blocks=10, filters=256):
    log.info("Loading dataset")
    data_train = DatasetGame()
    data_train.load(dataset_path, slot_free=True)
    os.makedirs(model_dir, exist_ok=True)
    model_path = get_model_path(model_dir)
    log.info("Loading the agent...")
    if os.path.exists(model_path):
        chess_agent = Agent(color=True, weights=model_path)
    else:
        log.warning("Model not found, training a fresh one.")
        chess_agent = Agent(color=True, blocks=blocks, filters=filters)
    history = chess_agent.train(data_train, logdir=model_dir, epochs=epochs, validation_split=0.25,
                                batch_size=batch_size)
    log.info("Saving the agent...")
    chess_agent.save(model_path)
    return history


def main(argv=None):
    parser = argparse.ArgumentParser(description="Trains a model on recorded games.")
    parser.add_argument("model_dir", metavar="modeldir",
                        help="where to store (and load from) the trained model and the logs")
    parser.add_argument("data_path", metavar="datadir", help="Path of .JSON dataset.")
    parser.add_argument("--epochs", type=int, default=1)
    parser.add_argument("--bs", type=int, default=8, help="Batch size (games). Default 8")
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--blocks", type=int, default=10)
    parser.add_argument("--filters", type=int, default=256)
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.INFO)
    train(args.model_dir, args.data_path, args.epochs, args.bs, args.blocks, args.filters)


if __name__ == "__main__":
    main()
