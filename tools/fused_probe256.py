"""Scratch probe (GPU): fused 256-filter trunk vs PyTorch trunk vs fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tower_oracle
from chessrl_amd.model import ChessModel
rng = np.random.default_rng(0)
planes = (rng.random((64, 8, 8, 127)) < 0.12).astype(np.float32)
for blocks in (1, 3, 20):
    for rbn in (False, True):
        w = tower_oracle.init_weights(blocks, 256, seed=4, randomize_bn=rbn)
        ep, ev = tower_oracle.forward(w, planes)
        for fused in (True, False):
            m = ChessModel(weights=w, fused=fused)
            kp, kv = m.predict(planes)
            print("%2dx256 rbn=%d fused=%d dv=%.2e dp=%.2e" % (blocks, rbn, fused,
                  np.abs(kv[:, 0] - ev.numpy()).max(), np.abs(kp - ep.numpy()).max()), flush=True)
B = 4096
x = torch.zeros((B, 8, 8, 128), dtype=torch.float16, device="cuda:0")
x[..., :127] = torch.from_numpy((rng.random((B, 8, 8, 127)) < 0.12).astype(np.float16)).cuda()
for blocks, filters in ((20, 256), (10, 256), (10, 128)):
    for fused in (True, False):
        m = ChessModel(blocks=blocks, filters=filters, fused=fused)
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m(x)
        e1.record(); torch.cuda.synchronize()
        d = e0.elapsed_time(e1) / 5
        print("%2dx%d B=%d fused=%d forward %.3f ms (%.0f TFLOP/s)" % (blocks, filters, B, fused, d, 2 * m.macs_per_eval() * B / d / 1e9), flush=True)
