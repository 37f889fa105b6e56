"""Scratch (GPU): host cost of a move boundary (end_move + begin_move) at C3 scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = ChessModel(blocks=10, filters=128)
run = SelfPlayRunner(model, G, 32, seed=0, noise=True)
import cProfile, pstats
for rep in range(3):
    run.begin_move()
    for _ in range(32):
        run.engine.step()
    run._sims_in_move = 32
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    run.end_move()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run.begin_move()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rep == 2:
        pr.disable()
    run._sims_in_move = None
    print("end_move %.1f ms  begin_move %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
