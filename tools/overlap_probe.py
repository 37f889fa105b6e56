"""Scratch (GPU): would two half-populations on two streams hide the search kernels and heads of one
half under the other half's trunk launches?  Two LockstepEngines of G/2 games,
each with its own hipGraph of one step, replayed alternately on two streams, against one engine of G.
python tools/overlap_probe.py [G=4096] [steps=400]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd.engine import LockstepEngine
from chessrl_amd.model import ChessModel

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
model = ChessModel(blocks=10, filters=128)


def prepared(n, stream):
    with torch.cuda.stream(stream):
        eng = LockstepEngine(model, n, 800)
        eng.reset()
        eng.search_begin()
        for _ in range(150):                 # mid-move depth
            eng.step()
    stream.synchronize()
    return eng


def timed(engs, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


s0 = torch.cuda.Stream()
one = prepared(G, s0)
t_one = timed([one], [s0])
one.close()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
a, b = prepared(G // 2, sa), prepared(G // 2, sb)
t_two = timed([a, b], [sa, sb])
t_seq = timed([a, b], [sa, sa])               # the same two halves on ONE stream (no overlap possible)
print("one engine of %d games: %.4f ms/step | two halves on two streams: %.4f | two halves on one stream: %.4f"
      % (G, t_one, t_two, t_seq))
