"""Scratch (GPU): a few launches of the production fused / layer-wise trunk from plane bitboards, for rocprofv3
passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_...`, `--kernel-trace --stats`).
python tools/trunk_once.py [blocks=10] [filters=128] [boards=4096] [precision=f16 | f16x3 | indexed]
`indexed` = the fall-back launch of the hybrid mode (crl_trunk_forward_indexed) with every 13th board listed
(7.7 %: what a C3 step lists at 2 x the probe distance)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd import _lib
from chessrl_amd.model import ChessModel
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
filters = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
mode = sys.argv[4] if len(sys.argv) > 4 else "f16"
rng = np.random.default_rng(0)
bits = torch.from_numpy(rng.integers(0, 1 << 62, (B, 128), dtype=np.int64) &
                        rng.integers(0, 1 << 62, (B, 128), dtype=np.int64) &
                        rng.integers(0, 1 << 62, (B, 128), dtype=np.int64)).cuda()     # ~12 % of the bits set
bits[:, 127] = 0
m = ChessModel(blocks=blocks, filters=filters, precision="f16x3" if mode == "indexed" else mode)
if mode == "indexed":
    vp = ctypes.c_void_p
    pick = torch.arange(0, B, 13, dtype=torch.int32)
    lst = torch.zeros(_lib.LIST_HEADER + B, dtype=torch.int32, device="cuda")
    lst[0] = len(pick)
    lst[_lib.LIST_HEADER:_lib.LIST_HEADER + len(pick)] = pick.cuda()
    hp = torch.zeros((B, 192), dtype=torch.float32, device="cuda")
    ws = m._trunk_workspace(B)
    for _ in range(3):
        rc = _lib.lib().crl_trunk_forward_indexed(
            vp(torch.cuda.current_stream().cuda_stream), filters, vp(bits.data_ptr()), vp(m._wtiles3.data_ptr()),
            vp(m._wbias.data_ptr()), B, blocks, vp(m._head_w.data_ptr()), vp(m._head_b.data_ptr()), vp(hp.data_ptr()),
            vp(lst.data_ptr()), vp(ws.data_ptr() if ws is not None else None), ws.numel() if ws is not None else 0)
        assert rc == 0
else:
    for _ in range(3):
        m._run_fused(bits)
torch.cuda.synchronize()
