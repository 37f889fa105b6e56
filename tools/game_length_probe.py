"""Scratch (GPU): play complete self-play games with the real random-init tower; ply distribution."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner
games, sims = int(sys.argv[1]), int(sys.argv[2])
model = ChessModel(blocks=10, filters=128)
run = SelfPlayRunner(model, games, sims, seed=0, noise=True, total_games=games, max_plies=4096)
t0 = time.time()
recs = run.run()
dt = time.time() - t0
pl = np.array([len(r) for r in recs])
res = np.array([r.result for r in recs])
out = {"games": len(recs), "sims_per_move": sims, "seconds": dt, "sims_run": run.sims_run,
       "sims_per_s": run.sims_run / dt, "plies_mean": float(pl.mean()), "plies_median": float(np.median(pl)),
       "plies_min": int(pl.min()), "plies_max": int(pl.max()),
       "plies_pct": [int(x) for x in np.percentile(pl, [10, 25, 50, 75, 90])],
       "results": {"white": int((res == 1).sum()), "black": int((res == -1).sum()), "draw": int((res == 0).sum())},
       "moves_per_game_mean": float(pl.mean() / 2)}
print(json.dumps(out))
