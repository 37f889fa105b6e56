"""chessrl_amd -- MI355X-native self-play MCTS simulation loop behind ChessRL's
selfplay.py / Agent / Game API surface.  The hot path lives in
libchessrl_hip.so (hand-written gfx950 HIP kernels, C-ABI in include/chessrl_hip.h);
this package is the thin Python host mirror of the reference's interface.
"""
__version__ = "0.1.0"
