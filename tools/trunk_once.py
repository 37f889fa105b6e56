"""Scratch (GPU): a few launches of the production fused trunk from plane bitboards, for rocprofv3
passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--kernel-trace --stats`).
python tools/trunk_once.py [blocks=10] [filters=128] [boards=4096] [precision=f16]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
filters = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
rng = np.random.default_rng(0)
bits = torch.from_numpy(rng.integers(0, 1 << 62, (B, 128), dtype=np.int64) &
                        rng.integers(0, 1 << 62, (B, 128), dtype=np.int64) &
                        rng.integers(0, 1 << 62, (B, 128), dtype=np.int64)).cuda()     # ~12 % of the bits set
bits[:, 127] = 0
m = ChessModel(blocks=blocks, filters=filters, precision=sys.argv[4] if len(sys.argv) > 4 else "f16")
for _ in range(3):
    m._run_fused(bits)
torch.cuda.synchronize()
