"""Scratch (GPU): HIP-event time of crl_heads_forward (policy + value, policy only) per batch size, in
the one-pass form and as label slices + normalising pass (scratch given), inside a hipGraph."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chessrl_amd import _lib
from chessrl_amd.model import ChessModel
m = ChessModel(blocks=2, filters=64, precision="f16")
vp = ctypes.c_void_p
_lib.lib().crl_heads_set_sliced_max(1 << 20)            # the form is chosen by the scratch argument here
for n in (64, 256, 512, 1024, 2048, 4096, 8192):
    hp = torch.rand((n, 192), device="cuda")
    pol = torch.empty((n, 1968), device="cuda"); val = torch.empty((n,), device="cuda")
    pri = torch.empty((n, 256), device="cuda"); scratch = torch.zeros((n, 16), device="cuda")
    cnt = torch.full((n,), 30, dtype=torch.int32, device="cuda")
    lab = (torch.arange(256, device="cuda")[None, :] * 7 + torch.arange(n, device="cuda")[:, None]).remainder(1968).to(torch.int16)
    def run(with_value, sliced, legal):
        sc = vp(scratch.data_ptr() if sliced else None)
        common = (vp(torch.cuda.current_stream().cuda_stream), vp(hp.data_ptr()), n,
                  vp(m._pol_wp.data_ptr()), vp(m._pol_bias.data_ptr()), vp(m._val_w1p.data_ptr()), vp(m._val_b1.data_ptr()),
                  vp(m._val_w2.data_ptr()))
        if legal:
            _lib.lib().crl_heads_forward_legal(*common, vp(lab.data_ptr()), vp(cnt.data_ptr()), vp(pri.data_ptr()),
                                               vp(val.data_ptr() if with_value else None), sc)
        else:
            _lib.lib().crl_heads_forward(*common, vp(pol.data_ptr()), vp(val.data_ptr() if with_value else None), sc)
    res = {}
    for sliced in (False, True):
        for legal in (True, False):
            for wv in (True, False):
                for _ in range(5): run(wv, sliced, legal)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()                      # inside a hipGraph, as the engine runs it:
                with torch.cuda.graph(g):                        # eager Python launches are host-bound
                    for _ in range(20): run(wv, sliced, legal)
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): g.replay()
                e1.record(); torch.cuda.synchronize()
                res[(sliced, legal, wv)] = e0.elapsed_time(e1) / 100 * 1e3
        run(True, sliced, False); torch.cuda.synchronize()
        ref = torch.softmax(m.net.policy_fc(hp[:, :128]), -1)
        res[("err", sliced)] = (pol - ref).abs().max().item()
    print("n=%5d  legal priors, policy+value / policy only: one-pass %.1f / %.1f us, sliced %.1f / %.1f us | full vectors: "
          "one-pass %.1f / %.1f, sliced %.1f / %.1f | max|dp| vs torch %.1e / %.1e" % (
              n, res[(False, True, True)], res[(False, True, False)], res[(True, True, True)], res[(True, True, False)],
              res[(False, False, True)], res[(False, False, False)], res[(True, False, True)], res[(True, False, False)],
              res[("err", False)], res[("err", True)]), flush=True)
