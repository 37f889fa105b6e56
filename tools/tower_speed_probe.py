"""Scratch probe (GPU): tower forward time by dtype / batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
rng = np.random.default_rng(0)
for blocks, filters in [(6, 64), (10, 128), (20, 256)]:
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        m = ChessModel(blocks=blocks, filters=filters, dtype=dt)
        for B in (512, 4096):
            x = torch.zeros((B, 8, 8, 128), dtype=dt, device="cuda:0")
            x[..., :127] = torch.from_numpy((rng.random((B, 8, 8, 127)) < 0.12).astype(np.float32)).cuda().to(dt)
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t = time.time(); n = 5
            for _ in range(n):
                m(x)
            torch.cuda.synchronize()
            d = (time.time() - t) / n
            print("%2dx%-3d %-8s B=%4d  %.2f ms  %.0f TFLOP/s" % (blocks, filters, str(dt)[6:], B, d * 1e3, 2 * m.macs_per_eval() * B / d / 1e12), flush=True)
