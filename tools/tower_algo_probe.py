"""Scratch probe (GPU): MIOpen solver choice vs error and speed of the fp16 tower."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tower_oracle
from chessrl_amd.model import ChessModel

blocks, filters, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
planes = (rng.random((64, 8, 8, 127)) < 0.12).astype(np.float32)
for rbn in (False, True):
    w = tower_oracle.init_weights(blocks, filters, seed=4, randomize_bn=rbn)
    ep, ev = tower_oracle.forward(w, planes)
    m = ChessModel(weights=w)
    kp, kv = m.predict(planes)
    print("env WINO=%s %dx%d rbn=%d dv=%.2e dp=%.2e" % (os.environ.get("MIOPEN_DEBUG_CONV_WINOGRAD"), blocks, filters, rbn,
          np.abs(kv[:, 0] - ev.numpy()).max(), np.abs(kp - ep.numpy()).max()), flush=True)
x = torch.zeros((B, 8, 8, 128), dtype=torch.float16, device="cuda:0")
x[..., :127] = torch.from_numpy((rng.random((B, 8, 8, 127)) < 0.12).astype(np.float16)).cuda()
for _ in range(3):
    m(x)
torch.cuda.synchronize()
t = time.time()
n = 10
for _ in range(n):
    m(x)
torch.cuda.synchronize()
dt = (time.time() - t) / n
macs = m.macs_per_eval()
print("B=%d forward %.3f ms  -> %.1f TFLOP/s" % (B, dt * 1e3, 2 * macs * B / dt / 1e12), flush=True)
