"""Scratch (GPU): host cost of a move boundary (end_move + begin_move), profiled; plain boundaries and
HARVEST boundaries (a game finished: records fetched, slot reset, greedy opening) apart.
python tools/boundary_probe.py [G=4096] [blocks=10] [filters=128] [sims=32] [moves=12] [max_plies=2048]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
filters = int(sys.argv[3]) if len(sys.argv) > 3 else 128
sims = int(sys.argv[4]) if len(sys.argv) > 4 else 32
moves = int(sys.argv[5]) if len(sys.argv) > 5 else 12
max_plies = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
model = ChessModel(blocks=blocks, filters=filters, precision="f16")
run = SelfPlayRunner(model, G, sims, seed=0, noise=True, max_plies=max_plies)
import cProfile, pstats
plain, harvest = [], []
prof = {"plain": cProfile.Profile(), "harvest": cProfile.Profile()}
for rep in range(moves):
    run.begin_move()
    for i in range(sims):
        run.engine.step()
        if i + 1 == max(1, sims // 2):
            run._sims_in_move = i + 1
            run._draw_noise_ahead()
    run._sims_in_move = sims
    torch.cuda.synchronize()
    n0 = len(run.finished)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    run.end_move()
    torch.cuda.synchronize()
    pr.disable()
    t1 = time.perf_counter()
    kind = "harvest" if len(run.finished) > n0 else "plain"
    (harvest if kind == "harvest" else plain).append(((t1 - t0) * 1e3, len(run.finished) - n0))
    if rep >= 2:
        last = {kind: pr}
        prof.update(last)
    run._sims_in_move = None
print("plain boundaries: %d, end_move mean %.2f ms (first two excluded: %.2f)" % (
    len(plain), np.mean([p[0] for p in plain]), np.mean([p[0] for p in plain[2:]] or [0])))
if harvest:
    print("harvest boundaries: %d, end_move mean %.2f ms, median %.2f, games per harvest %.1f" % (
        len(harvest), np.mean([h[0] for h in harvest]), np.median([h[0] for h in harvest]), np.mean([h[1] for h in harvest])))
    pstats.Stats(prof["harvest"]).sort_stats("cumulative").print_stats(25)
else:
    pstats.Stats(prof["plain"]).sort_stats("cumulative").print_stats(18)
