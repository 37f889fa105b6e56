"""Scratch (GPU): what the head-room of the hybrid mode's reply margin costs.  margin = HYBRID_K x the f16-vs-f16x3 distance
of the weights on the probe positions (model.py); a larger K lists more S1 boards for the split-precision fall-back launch.
For K in the list: fraction of S1 boards evaluated twice and ms per step of a hybrid step mid-move.
python tools/hybrid_k_probe.py [G=4096] [blocks=10] [filters=128] [seed=1] [K list=2,3,4] [sharp=0]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
filters = int(sys.argv[3]) if len(sys.argv) > 3 else 128
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ks = [float(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "2,3,4").split(",")]
sharp = len(sys.argv) > 6 and sys.argv[6] == "1"
weights = None
if sharp:
    from oracle import tower_oracle
    from tests.util import encode_prefixes, selfplay_position_prefixes
    prefixes, _ = selfplay_position_prefixes(512)
    _, planes = encode_prefixes(ChessModel(blocks=2, filters=64, precision="f16"), prefixes)
    weights = tower_oracle.calibrated_weights(blocks, filters, planes[:512], seed=7)
out = []
for k in ks:
    ChessModel.HYBRID_K = k
    model = ChessModel(blocks=blocks, filters=filters, seed=seed, precision="hybrid", weights=weights)
    run = SelfPlayRunner(model, G, 800, seed=seed, noise=True, max_plies=2048)
    run.step(); run.end_move()
    run.steps(408)
    run.engine.prepare_graphs(40)
    torch.cuda.synchronize()
    c0, fb0 = run.engine.ctx.counters()["sims"], model.fallback_boards()
    t0 = time.perf_counter()
    run.steps(80)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sims = run.engine.ctx.counters()["sims"] - c0
    out.append({"hybrid_k": k, "reply_margin": model.reply_margin, "ms_per_step": dt / 80 * 1e3,
                "simulations_per_s": sims / dt, "s1_boards_evaluated_twice": (model.fallback_boards() - fb0) / max(1, sims)})
    print(out[-1], flush=True)
    run.close()
print(json.dumps({"games": G, "tower": "%dx%d" % (blocks, filters), "seed": seed, "sharp": sharp, "rows": out}))
