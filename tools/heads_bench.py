"""Scratch (GPU): HIP-event time of crl_heads_forward (policy + value, policy only) per batch size."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd import _lib
from chessrl_amd.model import ChessModel
m = ChessModel(blocks=2, filters=64)
vp = ctypes.c_void_p
for n in (512, 1024, 2048, 4096, 8192):
    hp = torch.rand((n, 192), device="cuda")
    pol = torch.empty((n, 1968), device="cuda"); val = torch.empty((n,), device="cuda")
    def run(with_value):
        _lib.lib().crl_heads_forward(vp(torch.cuda.current_stream().cuda_stream), vp(hp.data_ptr()), n,
            vp(m._pol_wp.data_ptr()), vp(m._pol_bias.data_ptr()), vp(m._val_w1p.data_ptr()), vp(m._val_b1.data_ptr()),
            vp(m._val_w2.data_ptr()), vp(pol.data_ptr()), vp(val.data_ptr() if with_value else None))
    out = []
    for wv in (True, False):
        for _ in range(5): run(wv)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()                      # inside a hipGraph, as the engine runs it:
        with torch.cuda.graph(g):                        # eager Python launches are host-bound
            for _ in range(20): run(wv)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 100 * 1e3)
    ref = torch.softmax(m.net.policy_fc(hp[:, :128]), -1)
    print("n=%5d: policy+value %.1f us, policy only %.1f us; max|dp| vs torch %.2e, row sums %.6f..%.6f" % (
        n, out[0], out[1], (pol - ref).abs().max().item(), pol.sum(1).min().item(), pol.sum(1).max().item()))
