"""Scratch (GPU): run the fused trunk a few times (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
B = 4096
rng = np.random.default_rng(0)
x = torch.zeros((B, 8, 8, 128), dtype=torch.float16, device="cuda:0")
x[..., :127] = torch.from_numpy((rng.random((B, 8, 8, 127)) < 0.12).astype(np.float16)).cuda()
m = ChessModel(blocks=10, filters=128)
for _ in range(3):
    m._run_fused(x)
torch.cuda.synchronize()
