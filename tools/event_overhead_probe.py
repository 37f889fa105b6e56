"""Scratch (GPU): what HIP events between the phases of an eager step cost (bench.py's profile_phases), C5 hybrid.
python tools/event_overhead_probe.py [blocks=20] [filters=256] [G=4096] [steps=12]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chessrl_amd.model import ChessModel
from chessrl_amd.selfplay import SelfPlayRunner

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
filters = int(sys.argv[2]) if len(sys.argv) > 2 else 256
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
model = ChessModel(blocks=blocks, filters=filters, precision="hybrid")
run = SelfPlayRunner(model, G, 800, seed=0, noise=True, max_plies=2048, use_graph=False)
run.step(); run.end_move()
run.steps(200)
eng = run.engine
torch.cuda.synchronize()


def timed(label, body):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    print("%-60s %8.3f ms per step" % (label, (time.perf_counter() - t0) / n * 1e3), flush=True)


timed("plain eager steps", eng._step_body)
for name, cls in (("torch.cuda.Event", torch.cuda.Event), ("NoFenceEvent", bench.NoFenceEvent)):
    def phases():
        evs = [cls(enable_timing=True) for _ in range(5)]
        evs[0].record(); eng.phase_select_expand(); evs[1].record(); eng.phase_tower_s1(); evs[2].record()
        eng.phase_reply(); evs[3].record(); eng.phase_tower_s2(); evs[4].record()
        keep.append(evs)
    keep = []
    timed("5 phase events per step, %s" % name, phases)
    model.trunk_events, model.trunk_event_cls = [], cls
    timed("trunk events only (6 per step), %s" % name, eng._step_body)
    keep = []
    timed("phase + trunk events (11 per step), %s" % name, phases)
    model.trunk_events = None
pre = [[bench.NoFenceEvent() for _ in range(5)] for _ in range(n)]
it = iter(pre)


def phases_pre():
    evs = next(it)
    evs[0].record(); eng.phase_select_expand(); evs[1].record(); eng.phase_tower_s1(); evs[2].record()
    eng.phase_reply(); evs[3].record(); eng.phase_tower_s2(); evs[4].record()


timed("5 phase events per step, NoFenceEvent created beforehand", phases_pre)
run.close()
