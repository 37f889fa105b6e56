"""Scratch (GPU): time crl_trunk128_forward variants (CRL_TRUNK_VARIANT) at C3 shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
B, blocks = 4096, 10
rng = np.random.default_rng(0)
x = torch.zeros((B, 8, 8, 128), dtype=torch.float16, device="cuda:0")
x[..., :127] = torch.from_numpy((rng.random((B, 8, 8, 127)) < 0.12).astype(np.float16)).cuda()
m = ChessModel(blocks=blocks, filters=128)
flops = 2.0 * (73152 * 128 + 1152 * 128 * 128 * blocks + 192 * 128) * B
variants = [int(v) for v in sys.argv[1:]] or [0]
res = {}
for rnd in range(3):
    for v in variants:
        os.environ["CRL_TRUNK_VARIANT"] = str(v)
        for _ in range(2):
            m._run_fused(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m._run_fused(x)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
for v in variants:
    t = min(res[v])
    print("variant %3d: %.3f ms (min of 3) = %.0f TFLOP/s   all=%s" % (v, t, flops / t / 1e9, ["%.3f" % z for z in res[v]]))
