"""chessrl_amd -- MI355X-native self-play MCTS simulation loop behind ChessRL's
selfplay.py / Agent / Game API surface.  The hot path lives in
libchessrl_hip.so (hand-written gfx950 HIP kernels, C-ABI in include/chessrl_hip.h);
this package is the thin Python host mirror of the reference's interface.
"""
__version__ = "0.1.0"


def multiprocess_env():
    """Multi-process GPU work on this ROCm stack (RCCL communicators, device tensors shared between rank
    processes) needs dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0, read when HSA initialises, i.e. at the first GPU
    call.  Called by the multi-process entry points (``selfplay.main`` with WORLD_SIZE > 1; bench.py sets the same
    before it imports torch) -- NOT at package import: a host application that merely imports chessrl_amd keeps
    its own ROCm IPC behaviour.  An explicit setting in the caller's environment wins; when HSA is
    already up the variable can no longer take effect and a warning says so.  Returns the value in force."""
    import os
    import warnings
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
        try:
            import torch
            late = torch.cuda.is_initialized()
        except Exception:
            late = False
        if late:
            warnings.warn("chessrl_amd: the GPU was initialised before HSA_ENABLE_IPC_MODE_LEGACY=0 could be set; RCCL / "
                          "cross-process device tensors may fail with hipIpcGetMemHandle: invalid argument -- export it "
                          "in the environment of the launcher", RuntimeWarning)
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]
