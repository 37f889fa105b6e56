"""Scratch (GPU): do P sub-populations of G/P games on P streams -- each its own LockstepEngine and hipGraph of
one step, replayed round-robin -- hide each other's launch gaps and latency-bound kernels (SURVEY section 7
step 7)?  At C3 the trunk fills every CU and nothing overlaps (round 2: 2.345 vs 2.341 ms); at C2 every
kernel of the step is latency-bound on a fraction of the chip.
python tools/overlap_probe.py [G=4096] [steps=400] [blocks=10] [filters=128] [sims=800] [parts=1,2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd.engine import LockstepEngine
from chessrl_amd.model import ChessModel

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 10
filters = int(sys.argv[4]) if len(sys.argv) > 4 else 128
sims = int(sys.argv[5]) if len(sys.argv) > 5 else 800
parts = [int(x) for x in (sys.argv[6] if len(sys.argv) > 6 else "1,2").split(",")]
model = ChessModel(blocks=blocks, filters=filters, precision="f16")
grow = min(150, sims // 2)


def prepared(n, stream):
    with torch.cuda.stream(stream):
        eng = LockstepEngine(model, n, sims)
        eng.reset()
        eng.search_begin()
        for _ in range(grow):                # mid-move depth
            eng.step()
    stream.synchronize()
    return eng


def timed(engs, streams, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


n_timed = min(steps, sims - grow - 2)
for p in parts:
    streams = [torch.cuda.Stream() for _ in range(p)]
    engs = [prepared(G // p, s) for s in streams]
    t_par = timed(engs, streams, n_timed // 2)
    for e, s in zip(engs, streams):          # back to the same depth for the second measurement
        with torch.cuda.stream(s):
            e.search_begin()
            for _ in range(grow):
                e.step()
    t_seq = timed(engs, [streams[0]] * p, n_timed // 2) if p > 1 else float("nan")
    print("%d games as %d x %d on %d streams: %.4f ms per step of all games (%.3f M sims/s) | the same parts on ONE "
          "stream: %.4f" % (G, p, G // p, p, t_par, G / t_par / 1e3, t_seq), flush=True)
    for e in engs:
        e.close()
