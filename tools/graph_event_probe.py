"""Scratch (GPU), round 6: can HIP timing events be captured INTO a hipGraph on this stack (VERDICT r5 #1 route b)?
torch.cuda.Event(enable_timing=True, external=True) records as an event-record NODE under capture
(hipEventRecordWithFlags(..., hipEventRecordExternal)); after a replay elapsed_time between two such events should be
the time of the kernels between them.  Prints what happens; the product's bench uses crl_stamp kernels (which need
nothing from the runtime) whatever the answer.  python tools/graph_event_probe.py"""
import json
import sys
import torch

out = {"torch": torch.__version__, "hip": torch.version.hip}
try:
    x = torch.randn((4096, 4096), device="cuda", dtype=torch.float16)
    y = torch.empty_like(x)
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
    torch.matmul(x, x, out=y)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e0.record()
        for _ in range(10):
            torch.matmul(x, x, out=y)
        e1.record()
    times = []
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    # the same ten products between ordinary events, eagerly
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        torch.matmul(x, x, out=y)
    b.record()
    torch.cuda.synchronize()
    out.update(external_events_in_graph_ms=times, eager_events_ms=a.elapsed_time(b), works=True)
except Exception as e:                                   # noqa: BLE001 -- the probe reports whatever the stack says
    out.update(works=False, error="%s: %s" % (type(e).__name__, str(e)[:500]))
print(json.dumps(out))
