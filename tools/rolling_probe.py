"""Scratch (GPU): self-play games per hour over whole games, stop-and-train rounds against rolling rounds
(SelfPlayRunner.run_rolling) at C3: G games in lockstep, S sims/move, 10x128 random-init tower.
python tools/rolling_probe.py [G=4096] [S=800] [rounds=3] [round_size=G] [train | notrain] [share=0.2] [tower=10x128]
Reports when each round was handed over and, from the timeline of finished games, the rate over every
window of `round_size` consecutively finished games (the accounting window).  With `train` every round is
handed to the background trainer (chessrl_amd.selfplay.BackgroundTrainer: rank 0's thread and stream) as the
CLI's --rolling mode does, and the weights are swapped in place when a weight set is ready: games per hour
WITH training in the loop."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 800
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N = int(sys.argv[4]) if len(sys.argv) > 4 else G
TRAIN = len(sys.argv) > 5 and sys.argv[5] == "train"
SHARE = float(sys.argv[6]) if len(sys.argv) > 6 else 0.2      # of the wall time, to the trainer while it has work
TOWER = [int(x) for x in (sys.argv[7] if len(sys.argv) > 7 else "10x128").split("x")]
# precision "auto" (what the product runs): the probe decides, and the run-time guard of an auto-kept f16 re-checks it on the
# run's own tree leaves every 8 move boundaries (rounds 3-4 ran the no-training probe in --precision f16)
# CRL_PRECISION / CRL_SEED: another mode or another random-init net (seed 1 is one that auto runs in hybrid)
model = ChessModel(blocks=TOWER[0], filters=TOWER[1], precision=os.environ.get("CRL_PRECISION", "auto"),
                   seed=int(os.environ.get("CRL_SEED", "0")))
START_PRECISION = model.precision
run = SelfPlayRunner(model, G, S, seed=0, noise=True, total_games=R * N, round_size=N, max_plies=4096)
t0 = time.time()
timeline = []                                   # (seconds, games finished so far, slots in the batch)
taken = [0]
orig = run.play_move


def play_move():
    out = orig()
    timeline.append((time.time() - t0, taken[0] + len(run.finished), run.G))
    return out


run.play_move = play_move
rounds = []


def on_round(r, recs):
    taken[0] += len(recs)
    pl = np.array([len(x) for x in recs])
    rounds.append({"round": r, "handed_over_at_s": time.time() - t0, "games": len(recs),
                   "plies_mean": float(pl.mean()), "plies_max": int(pl.max())})
    print(json.dumps(rounds[-1]), flush=True)


bg, loads = None, []
if TRAIN:
    from chessrl_amd.selfplay import BackgroundTrainer
    bg = BackgroundTrainer(model.weights, "cuda:0")
    on_round_plain = on_round

    def on_round(r, recs):
        on_round_plain(r, recs)
        bg.submit(r, recs)

    def on_news(k):
        model.load_dict(bg.latest()[0])
        loads.append({"weight_set": k, "at_s": time.time() - t0, "precision": model.precision})
        print(json.dumps(loads[-1]), flush=True)

    idle = (lambda dt: time.sleep(dt * SHARE / (1.0 - SHARE)) if bg.busy() else None) if SHARE > 0 else None
    done = run.run_rolling(R, on_round=on_round, poll=bg.ready, on_news=on_news, idle=idle)
else:
    done = run.run_rolling(R, on_round=on_round)
total = time.time() - t0
if bg is not None:
    bg.drain()
    trainer = [{"round": r, "seconds": s, "last": h} for r, s, h in bg.latest()[1]]
    bg.close()
tl = np.array(timeline)
# the rate over every window of N consecutively finished games: first time the count reaches k and k + N
windows = []
for k in range(0, int(tl[-1, 1]) - N + 1, max(1, N // 4)):
    ta = tl[np.searchsorted(tl[:, 1], k, side="left"), 0] if k > 0 else 0.0
    tb = tl[np.searchsorted(tl[:, 1], k + N, side="left"), 0]
    windows.append({"from_game": k, "seconds": float(tb - ta), "games_per_hour": N / (tb - ta) * 3600.0})
out = {"games_in_lockstep": G, "sims_per_move": S, "tower": "%dx%d %s" % (TOWER[0], TOWER[1], model.precision),
       "tower_precision": model.precision, "tower_precision_at_start": START_PRECISION, "tower_precision_guard": model.guard, "round_size": N, "rounds": rounds,
       "training_in_the_loop": TRAIN, "trainer_share": SHARE if TRAIN else None, "weight_loads": loads, "trainer": trainer if TRAIN else None,
       "seconds_until_last_round_trained": (time.time() - t0) if TRAIN else None,
       "rounds_done": done, "seconds_total": total, "games_total": int(tl[-1, 1]), "sims_run": run.sims_run,
       "sims_per_s_overall": run.sims_run / total, "games_per_hour_overall": tl[-1, 1] / total * 3600.0,
       "windows_of_round_size_finished_games": windows,
       "best_window_games_per_hour": max(w["games_per_hour"] for w in windows),
       "note": "overall includes the start-up (every game young) and the last round's thinning tail; windows "
               "that lie inside the refilled phase show the sustained rate"}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/rolling_probe%s%s.json" % ("_train" if TRAIN else "", os.environ.get("CRL_TAG", "")), "w"), indent=1)
