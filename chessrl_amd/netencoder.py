"""Board encoder and move-label table -- host mirror of the reference's ``netencoder``.

``get_uci_labels`` (/root/reference/src/chessrl/netencoder.py:94-134) comes from the
library's own generator (crl_uci_label_moves); ``get_game_state``
(netencoder.py:72-91) runs the encoder KERNEL on the game's arena slot and copies the
planes back -- the batched engine never does that copy, it feeds the fp16 NHWC buffer
to the tower in place.  ``DataGameSequence`` (training generator) is out of scope.
"""
import numpy as np

from . import _lib
from .game import arena, move_to_uci

_labels = None


def get_uci_labels():
    """The 1968 UCI move strings, in the reference's order."""
    global _labels
    if _labels is None:
        _labels = [move_to_uci(m) for m in _lib.uci_label_moves()]
    return list(_labels)


def get_game_state(game, flipped=False):
    """(8,8,127) float64 planes of ``game`` with its 8-ply history (netencoder.py:72-91)."""
    a = arena()
    planes = a.planes()
    ctx = a.one(game._slot)
    import torch
    ctx.set_stream(torch.cuda.current_stream(planes.device).cuda_stream)
    ctx.encode(planes.data_ptr())
    ctx.sync()
    current = planes[0, :, :, :127].float().cpu().numpy().astype(np.float64)
    if flipped:
        current = np.rot90(current, k=2)
    return current
