"""chessrl_amd -- MI355X-native self-play MCTS simulation loop behind ChessRL's
selfplay.py / Agent / Game API surface.  The hot path lives in
libchessrl_hip.so (hand-written gfx950 HIP kernels, C-ABI in include/chessrl_hip.h);
this package is the thin Python host mirror of the reference's interface.
"""
__version__ = "0.1.0"

import os as _os

# Multi-process GPU work on this ROCm stack (RCCL communicators, device tensors shared between rank processes)
# needs dmabuf IPC; the variable is read when HSA initialises, i.e. at the first GPU call, so setting it at
# package import is early enough.  An explicit setting in the caller's environment wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
