"""Scratch (GPU): how fast does the background trainer train BESIDE a full lockstep batch, and what does it cost
the self-play?  256 recorded games (played first, 16 sims/move) are trained by chessrl_amd.selfplay.BackgroundTrainer
while the main thread keeps a C3 batch (4096 games, 800 sims/move, 10x128) stepping; trainer seconds and self-play
simulations/s, with the trainer's stream at normal and at high priority, against each of them alone.
python tools/trainer_share_probe.py [games_to_train=256] [seconds=40]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
from chessrl_amd import selfplay
from chessrl_amd.selfplay import BackgroundTrainer, SelfPlayRunner

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
model = ChessModel(blocks=10, filters=128, seed=0, precision="f16")
side = SelfPlayRunner(model, n_train, 16, seed=9, noise=True, total_games=n_train, max_plies=1024)
recs = side.run()
side.close()
positions = sum(len(r.moves) for r in recs)
print("recorded %d games, %d positions" % (len(recs), positions), flush=True)
run = SelfPlayRunner(model, 4096, 800, seed=0, noise=True, max_plies=2048)
run.steps(50)
torch.cuda.synchronize()
out = {"games_trained": len(recs), "positions": positions}


def play(for_seconds, until=None):
    c0, t0 = run.engine.ctx.counters()["sims"], time.perf_counter()
    while (time.perf_counter() - t0 < for_seconds) if until is None else not until():
        run.steps(8)
        run.engine.ctx.sync()
    dt = time.perf_counter() - t0
    return (run.engine.ctx.counters()["sims"] - c0) / dt, dt


rate, _ = play(8)
out["selfplay_alone_sims_per_s"] = rate
bg = BackgroundTrainer(model.weights, "cuda:0")
bg.submit(0, recs)
bg.drain()
out["trainer_alone_s"] = bg.latest()[1][-1][1]
bg.close()
for prio in ("normal", "high"):
    selfplay.TRAINER_STREAM_PRIORITY = 0 if prio == "normal" else -1
    bg = BackgroundTrainer(model.weights, "cuda:0")
    bg.submit(0, recs)
    rate, dt = play(None, until=lambda: bg.ready() >= 1 or False)
    out["trainer_beside_selfplay_%s_priority_s" % prio] = bg.latest()[1][-1][1]
    out["selfplay_beside_trainer_%s_priority_sims_per_s" % prio] = rate
    bg.close()
    print(json.dumps(out), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/trainer_share_probe.json", "w"), indent=1)
