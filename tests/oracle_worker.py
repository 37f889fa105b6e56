"""Child process of tests.util.oracle_games_parallel: plays complete oracle self-play games on the
CPU.  stdin: JSON list of jobs; stdout: JSON {game id: Game.get_history()}."""
import json
import sys

import numpy as np


def main():
    from oracle import mcts_oracle
    from oracle.fakenet import FakeNet
    out = {}
    for j in json.load(sys.stdin):
        g = mcts_oracle.play_game(mcts_oracle.OracleAgent(FakeNet(**j["net"])), max_iters=j["sims"],
                                  noise=True, player_color=j["color"],
                                  rng=np.random.default_rng([j["seed"], j["gid"]]))
        out[j["gid"]] = g.get_history()
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()
