"""Scratch (GPU), round 6: at how many kernel nodes per hipGraph does `rocprofv3 --kernel-trace` stop surviving
hipGraphLaunch on this stack (ROCm 7.2)?  C2 / C3 benches trace fine (48 nodes per eight-step graph); a C5 hybrid step is
~135 nodes and segfaulted inside CUDAGraph::replay() even at --steps-per-graph 1 (profiles/r06).  Captures graphs of N
crl_stamp kernels, N rising, replays each three times and reports to stderr BEFORE and AFTER each, so the last line says
where the tracer died.  Run as:  rocprofv3 --kernel-trace --stats -d /tmp/x -- python3 tools/graph_trace_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd.engine import StampRing

ring = StampRing(1 << 16, "cuda:0")
for n in [int(x) for x in (sys.argv[1:] or "16 32 48 64 80 96 112 128 160 192 256 384 512 1024".split())]:
    print("graph of %d kernel nodes: capturing" % n, file=sys.stderr, flush=True)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(n):
            ring.stamp(i & 7)
    print("graph of %d kernel nodes: replaying" % n, file=sys.stderr, flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("graph of %d kernel nodes: ok" % n, file=sys.stderr, flush=True)
print("all graph sizes survived", file=sys.stderr, flush=True)
