"""Throughput of the training step: positions/s for one-game batches.

python tools/train_bench.py [--blocks 10 --filters 128 --plies 380 --steps 20]
Synthetic: random 0/1 planes of the encoder's density, random labels; batch sizes jitter around
--plies like real games do (every batch a different size).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from chessrl_amd.model import init_weights
from chessrl_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=10)
ap.add_argument("--filters", type=int, default=128)
ap.add_argument("--plies", type=int, default=380)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
tr = Trainer(init_weights(a.blocks, a.filters, seed=0), "cuda:0")
rng = np.random.default_rng(0)
sizes = [int(a.plies + rng.integers(-40, 41)) for _ in range(a.steps + 3)]
batches = []
for n in sizes:
    x = torch.zeros((n, 8, 8, 128), dtype=torch.float16, device="cuda:0")
    x[..., :127] = (torch.rand((n, 8, 8, 127), device="cuda:0") < 0.1).half()
    batches.append((x, torch.randint(0, 1968, (n,), device="cuda:0"),
                    torch.randint(-1, 2, (n,), device="cuda:0").float()))
for b in batches[:3]:
    tr.train_on_batch(*b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches[3:]:
    logs = tr.train_on_batch(*b)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
npos = sum(sizes[3:])
macs = 73152 * a.filters + 1152 * a.filters ** 2 * a.blocks + 192 * a.filters + 268544
print({"tower": "%dx%d" % (a.blocks, a.filters), "positions_per_s": npos / dt, "ms_per_game_batch": dt / a.steps * 1e3,
       "fp32_tflops": 3 * 2 * macs * npos / dt / 1e12, "loss": logs["loss"]})
