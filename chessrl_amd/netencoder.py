"""Board encoder and move-label table -- host mirror of the reference's ``netencoder``.

``get_uci_labels`` (/root/reference/src/chessrl/netencoder.py:94-134) comes from the
library's own generator (crl_uci_label_moves); ``get_game_state``
(netencoder.py:72-91) runs the encoder KERNEL on the game's arena slot and copies the
planes back -- the batched engine never does that copy, it feeds the fp16 NHWC buffer
to the tower in place.  ``DataGameSequence`` (netencoder.py:137-181, SURVEY.md section 8 row
f2) turns recorded games into training batches: all positions of a game are replayed in
lockstep through the push kernel and encoded by ONE launch of the encoder kernel.
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


class DataGameSequence(object):
    """Training-batch generator over a ``DatasetGame`` (netencoder.py:137-181).

    One batch = the augmented positions (dataset.py:21-43: one sample per move, target = the move
    played and the game's final result) of ``batch_size`` games; with probability ``random_flips``
    a game's planes are rotated by 180 degrees (``np.rot90(k=2)``, netencoder.py:89-90 -- the move
    label is NOT rotated, as in the reference).  ``__getitem__`` returns the reference's numpy
    triple; ``device_batch`` returns the same batch device-resident for the trainer
    (fp16 NHWC planes straight from the encoder kernel, label indices, results).
    """

    def __init__(self, dataset, batch_size=8, random_flips=0):
        self.dataset = dataset
        self.batch_size = min(batch_size, len(dataset))
        self.uci_ids = {u: i for i, u in enumerate(get_uci_labels())}
        self.random_flips = random_flips
        self._ctx = None

    def __len__(self):
        return int(len(self.dataset) / self.batch_size) if self.batch_size else 0

    def _context(self, n, device=0):
        if self._ctx is None or self._ctx.G < n or self._ctx.device != device:
            if self._ctx is not None:
                self._ctx.close()
            cap = (n + 63) // 64 * 64          # a sample sits at most n - 1 <= cap plies deep
            self._ctx = _lib.Context(max_games=cap, max_sims=1, max_plies=cap + 8, device=device)
        self._ctx.set_window(0, self._ctx.n_slots)
        return self._ctx

    def device_batch(self, idx, device="cuda:0"):
        """(planes f16 [N,8,8,128], move index i64 [N], result f32 [N]) on the GPU."""
        import torch
        from .game import uci_to_move
        games = self.dataset[idx * self.batch_size:(idx + 1) * self.batch_size]
        labels, results, flips, seqs, depth = [], [], [], [], []
        for g in games:
            hist = g.get_history()
            moves = hist["moves"]
            if moves and hist["result"] is None:
                raise ValueError("unfinished game in the training set (result is None)")
            flip = np.random.rand() < self.random_flips            # one draw per game
            ids = np.array([uci_to_move(m) for m in moves], dtype=np.uint16)
            for i, m in enumerate(moves):                          # sample i = position after i plies
                labels.append(self.uci_ids[m])
                results.append(hist["result"])
                flips.append(flip)
                seqs.append(ids)
                depth.append(i)
        n = len(labels)
        dev = torch.device(device)
        if n == 0:
            return (torch.zeros((0, 8, 8, _lib.PLANES), dtype=torch.float16, device=dev),
                    torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, device=dev))
        ctx = self._context(n, dev.index or 0)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.reset_games()
        depth = np.array(depth, dtype=np.int32)
        table = np.full((ctx.G, max(1, int(depth.max()))), _lib.NO_MOVE, dtype=np.uint16)
        for s in range(n):
            table[s, :depth[s]] = seqs[s][:depth[s]]
        counts = np.zeros(ctx.G, dtype=np.int32)
        counts[:n] = depth
        pushed = ctx.push_sequences(table, counts)               # one launch replays every prefix
        if not (pushed[:n] == depth).all():
            bad = int(np.argmax(pushed[:n] != depth))
            raise ValueError("recorded game holds an illegal move at ply %d" % int(pushed[bad]))
        planes = torch.empty((ctx.G, 8, 8, _lib.PLANES), dtype=torch.float16, device=dev)
        ctx.encode(planes.data_ptr())
        ctx.sync()
        planes = planes[:n]
        fl = torch.tensor(flips, device=dev)
        if bool(fl.any()):
            planes = torch.where(fl.view(-1, 1, 1, 1), planes.flip(1, 2), planes)
        return (planes.contiguous(), torch.tensor(labels, dtype=torch.int64, device=dev),
                torch.tensor(results, dtype=torch.float32, device=dev))

    def __getitem__(self, idx):
        planes, labels, results = self.device_batch(idx)
        x = planes[..., :127].float().cpu().numpy().astype(np.float64)
        y_pol = np.zeros((len(labels), 1968), dtype=np.float32)
        y_pol[np.arange(len(labels)), labels.cpu().numpy()] = 1.0
        return x, (y_pol, np.asarray(results.cpu().numpy(), dtype=np.float64))
