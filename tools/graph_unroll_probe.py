"""Scratch (GPU): does capturing K lockstep steps into ONE hipGraph (instead of one step per graph) save the
gap between consecutive graph launches?  python tools/graph_unroll_probe.py [G=512] [blocks=6] [filters=64] [sims=100]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd.engine import LockstepEngine
from chessrl_amd.model import ChessModel
G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
filters = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sims = int(sys.argv[4]) if len(sys.argv) > 4 else 100
model = ChessModel(blocks=blocks, filters=filters, precision="f16")
for K in (1, 2, 4, 8, 16):
    eng = LockstepEngine(model, G, sims)
    eng.STEPS_PER_GRAPH = K
    eng.reset()
    eng.search_begin()
    n = (sims - 2 * K) // K * K
    eng.run_steps(K)                            # capture + K steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_steps(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("K=%d steps per graph: %.2f us per step (%.3f M sims/s)" % (K, dt * 1e6, G / dt / 1e6), flush=True)
    eng.close()
