"""Scratch (GPU), round 6: does splitting the lockstep batch into P independent sub-populations on P HIP streams pay in the
HYBRID mode?  For plain f16 it does not (tools/overlap_probe.py: every kernel of the step fills the chip).  A hybrid
step, though, holds a launch sequence that does NOT fill it: the indexed f16x3 fall-back of the ~1 % of S1 boards whose
reply is a close call -- 41 latency-bound convolutions on a few dozen CUs, 2.6 ms of a 37-ms C5 step (7 %) with 200+
CUs idle -- and ~130 kernel boundaries per step.  With two sub-populations half a step apart, one's fall-back tower
runs beside the other's full-chip launches.
Every part has its own ChessModel (same seed: same weights; the model keeps per-batch scratch -- the fall-back list,
the layer-wise workspace -- so one model must not serve two streams), its own SelfPlayRunner (game ids interleaved:
rank p of P), stream and hipGraphs.  Games are put on lines of their own first (a few shortened noisy moves).
python tools/overlap_hybrid_probe.py [G=4096] [blocks=20] [filters=256] [sims=800] [parts=1,2] [steps=48] [precision=hybrid]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
filters = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sims = int(sys.argv[4]) if len(sys.argv) > 4 else 800
parts = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "1,2").split(",")]
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 48
precision = sys.argv[7] if len(sys.argv) > 7 else "hybrid"
K = 8
out = {"games": G, "tower": "%dx%d" % (blocks, filters), "sims": sims, "precision": precision, "steps_timed": steps, "runs": []}


def prepared(p, P, stream):
    with torch.cuda.stream(stream):
        model = ChessModel(blocks=blocks, filters=filters, seed=0, precision=precision)
        run = SelfPlayRunner(model, G // P, sims, seed=0, noise=True, rank=p, world=P, max_plies=1024)
        run.GUARD_EVERY = 0
        for _ in range(6):                               # noisy opening moves: every game on its own line
            run.begin_move()
            run.engine.run_steps(32)
            run._sims_in_move = 32
            run.end_move()
        run.begin_move()
        run.engine.run_steps(sims // 2)                  # mid-move trees
        run.engine.prepare_graphs(K)
        run.engine.run_steps(K)
    stream.synchronize()
    return run


def timed(runs, streams, n, offset):
    """n steps of every part; with `offset` the parts start half a step apart (part p > 0 gets a head start of K/2... a
    whole graph launch is K steps, so the stagger comes from launching part 0 first and letting the streams drift)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // K):
        for r, s in zip(runs, streams):
            with torch.cuda.stream(s):
                r.engine.run_steps(K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for P in parts:
    streams = [torch.cuda.Stream() for _ in range(P)]
    runs = [prepared(p, P, s) for p, s in enumerate(streams)]
    fb0 = [r.engine.evaluator.fallback_boards() for r in runs]
    c0 = [r.engine.ctx.counters()["sims"] for r in runs]
    ms = timed(runs, streams, steps, True)
    nsim = sum(r.engine.ctx.counters()["sims"] - c for r, c in zip(runs, c0))
    twice = sum(r.engine.evaluator.fallback_boards() - f for r, f in zip(runs, fb0)) / max(1, nsim)
    seq = None
    if P > 1:                                            # the same parts, same graphs, on ONE stream (no overlap)
        ms_seq = timed(runs, [streams[0]] * P, steps, False)
        seq = ms_seq
    e = {"parts": P, "games_per_part": G // P, "ms_per_step_of_all_games": ms, "simulations_per_s": G / ms * 1e3,
         "s1_boards_evaluated_twice": twice, "same_parts_on_one_stream_ms": seq, "mode": runs[0].engine.evaluator.precision}
    out["runs"].append(e)
    print(json.dumps(e), flush=True)
    for r in runs:
        r.close()
    del runs
    torch.cuda.empty_cache()
os.makedirs("gpurun_out/r06", exist_ok=True)
json.dump(out, open("gpurun_out/r06/overlap_hybrid_probe_%dx%d_%s.json" % (blocks, filters, precision), "w"), indent=1)
