"""Scratch (GPU): whole self-play games (noise on, refill, compaction) against the CPU oracle,
more of them than the test suite plays.  python tools/soak_parity.py [games=48] [sims=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chessrl_amd.selfplay import SelfPlayRunner, game_color
from oracle import mcts_oracle
from oracle.fakenet import FakeNet

n, sims = (int(sys.argv[1]) if len(sys.argv) > 1 else 48), (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
seed = 31
net = FakeNet(seed=77, prior_shift=30)
t0 = time.time()
run = SelfPlayRunner(net.to("cuda:0"), n_parallel=128, sims=sims, seed=seed, noise=True, total_games=n)
recs = {r.game_id: r for r in run.run()}
print("GPU: %d games in %.1fs, final batch %d slots" % (len(recs), time.time() - t0, run.G))
bad = 0
t0 = time.time()
for gid in sorted(recs):
    g = mcts_oracle.play_game(mcts_oracle.OracleAgent(net), max_iters=sims, noise=True,
                              player_color=game_color(seed, gid), rng=np.random.default_rng([seed, gid]))
    h = g.get_history()
    ok = recs[gid].get_history()["moves"] == h["moves"] and recs[gid].result == h["result"]
    bad += not ok
    if not ok:
        print("MISMATCH game", gid)
print("oracle: %.1fs; %d games, %d plies total, mismatches: %d" % (
    time.time() - t0, len(recs), sum(len(r) for r in recs.values()), bad))
sys.exit(1 if bad else 0)
