"""Finished-game records and their gather across GPUs.

The record is what the reference's ``Game.get_history()`` returns
(/root/reference/src/chessrl/game.py:59-66) and what ``DatasetGame.__str__``
serialises (dataset.py:80-82): ``{moves: [uci...], result, player_color, date}``.

Multi-GPU: games are independent, so there is no collective on the simulation path.
Only finished records travel: each rank packs them into one fixed-width int32 tensor
and one ``all_gather`` (RCCL over xGMI when the backend is ``nccl``; ``gloo`` in the
CPU tests) delivers every rank's rows to every rank.
"""
import json

import numpy as np

from .game import move_to_uci, uci_to_move

HEADER = 5      # game_id_lo, game_id_hi, plies, result, player_color


class GameRecord(object):
    __slots__ = ("game_id", "moves", "result", "player_color", "date")

    def __init__(self, game_id, moves, result, player_color, date=None):
        self.game_id = int(game_id)
        self.moves = np.asarray(moves, dtype=np.uint16)
        self.result = None if result is None else int(result)
        self.player_color = bool(player_color)
        self.date = date

    def get_history(self):
        return {"moves": [move_to_uci(m) for m in self.moves], "result": self.result,
                "player_color": self.player_color, "date": self.date}

    def __len__(self):
        return len(self.moves)

    def __eq__(self, other):
        return (self.game_id == other.game_id and self.result == other.result and
                self.player_color == other.player_color and np.array_equal(self.moves, other.moves))


def dumps(records):
    """JSON text in the layout of ``str(DatasetGame)`` (dataset.py:80-82)."""
    return json.dumps([r.get_history() for r in records])


def loads(text):
    out = []
    for i, item in enumerate(json.loads(text)):
        out.append(GameRecord(i, [uci_to_move(m) for m in item["moves"]], item["result"],
                              item["player_color"], item.get("date")))
    return out


def pack(records, max_plies):
    """records -> int32 [n, HEADER + ceil(max_plies/2)] (two u16 moves per int32)."""
    w = HEADER + (max_plies + 1) // 2
    out = np.zeros((len(records), w), dtype=np.int32)
    for i, r in enumerate(records):
        if len(r.moves) > max_plies:
            raise ValueError("record longer than max_plies")
        out[i, 0] = r.game_id & 0x7FFFFFFF
        out[i, 1] = r.game_id >> 31
        out[i, 2] = len(r.moves)
        out[i, 3] = 2 if r.result is None else r.result
        out[i, 4] = int(r.player_color)
        mv = np.zeros(2 * (w - HEADER), dtype=np.uint16)
        mv[:len(r.moves)] = r.moves
        out[i, HEADER:] = mv.view(np.int32)
    return out


def unpack(rows):
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = []
    for row in rows:
        n = int(row[2])
        mv = row[HEADER:].copy().view(np.uint16)[:n]
        res = None if row[3] == 2 else int(row[3])
        out.append(GameRecord(int(row[0]) | (int(row[1]) << 31), mv, res, bool(row[4])))
    return out


def gather_records(records, max_plies, device=None):
    """All ranks' finished records on every rank, ordered by game id.

    One all_gather of the per-rank counts and one of the padded record block.  Without an
    initialised process group (single GPU) it is the identity.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(records, key=lambda r: r.game_id)
    world = dist.get_world_size()
    dev = device if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl"
        else torch.device("cpu"))
    mine = torch.from_numpy(pack(records, max_plies)).to(dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev))
    nmax = max(int(c.item()) for c in counts)
    padded = torch.zeros((nmax, mine.shape[1]), dtype=torch.int32, device=dev)
    padded[:mine.shape[0]] = mine
    blocks = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(blocks, padded)
    out = []
    for c, b in zip(counts, blocks):
        out.extend(unpack(b[:int(c.item())].cpu().numpy()))
    return sorted(out, key=lambda r: r.game_id)
