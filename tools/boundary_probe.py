"""Scratch (GPU): host cost of a move boundary (end_move + begin_move), profiled.
python tools/boundary_probe.py [G=4096] [blocks=10] [filters=128] [sims=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
filters = int(sys.argv[3]) if len(sys.argv) > 3 else 128
sims = int(sys.argv[4]) if len(sys.argv) > 4 else 32
model = ChessModel(blocks=blocks, filters=filters, precision="f16")
run = SelfPlayRunner(model, G, sims, seed=0, noise=True)
import cProfile, pstats
for rep in range(4):
    run.begin_move()
    for i in range(sims):
        run.engine.step()
        if i + 1 == max(1, sims // 2):
            run._sims_in_move = i + 1
            run._draw_noise_ahead()
    run._sims_in_move = sims
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if rep == 3:
        pr = cProfile.Profile(); pr.enable()
    run.end_move()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run.begin_move()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rep == 3:
        pr.disable()
    run._sims_in_move = None
    print("end_move %.2f ms  begin_move %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
