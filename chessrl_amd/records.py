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
RESULT_NONE_WIRE = 2        # result None on the wire (the game was still running: Game.get_result(), game.py:92-109)
RESULT_TRUNCATED_WIRE = 3   # result None because the record reached max_plies and the runner ended the game


class GameRecord(object):
    __slots__ = ("game_id", "moves", "result", "player_color", "date", "truncated")

    def __init__(self, game_id, moves, result, player_color, date=None, truncated=False):
        self.game_id = int(game_id)
        self.moves = np.asarray(moves, dtype=np.uint16)
        self.result = None if result is None else int(result)
        self.player_color = bool(player_color)
        self.date = date
        # the game did not end by the rules: its record filled the engine's max_plies and the runner handed
        # it over as it stood (result None, as Game.get_result() says of a running game).  Not part of
        # get_history(): the reference's record has no such key.
        self.truncated = bool(truncated)

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


def row_width(max_plies):
    return HEADER + (max_plies + 1) // 2


def pack_arrays(game_ids, moves, plies, results, colors, max_plies):
    """The wire block straight from the device's record arrays (``crl_records``): int32
    [n, HEADER + ceil(max_plies/2)], two u16 moves per int32.  ``results`` uses 2 for None."""
    game_ids = np.asarray(game_ids, dtype=np.int64)
    plies = np.asarray(plies, dtype=np.int64)
    n, w = len(game_ids), row_width(max_plies)
    if n and plies.max() > max_plies:
        raise ValueError("record longer than max_plies")
    out = np.zeros((n, w), dtype=np.int32)
    out[:, 0] = game_ids & 0x7FFFFFFF
    out[:, 1] = game_ids >> 31
    out[:, 2] = plies
    out[:, 3] = np.asarray(results, dtype=np.int64)
    out[:, 4] = np.asarray(colors, dtype=np.int64)
    mv = np.zeros((n, 2 * (w - HEADER)), dtype=np.uint16)
    src = (np.asarray(moves, dtype=np.uint16).reshape(n, -1) if n else np.zeros((0, 0), np.uint16))[:, :max_plies]
    mv[:, :src.shape[1]] = np.where(np.arange(src.shape[1])[None, :] < plies[:, None], src, 0)
    out[:, HEADER:] = mv.view(np.int32)
    return out


def pack(records, max_plies):
    """records -> the wire block (see ``pack_arrays``)."""
    n = len(records)
    moves = np.zeros((n, max_plies), dtype=np.uint16)
    for i, r in enumerate(records):
        if len(r.moves) > max_plies:
            raise ValueError("record longer than max_plies")
        moves[i, :len(r.moves)] = r.moves
    return pack_arrays([r.game_id for r in records], moves, [len(r.moves) for r in records],
                       [(RESULT_TRUNCATED_WIRE if r.truncated else RESULT_NONE_WIRE) if r.result is None else r.result
                        for r in records],
                       [int(r.player_color) for r in records], max_plies)


def unpack(rows):
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    mv = rows[:, HEADER:].copy().view(np.uint16) if len(rows) else np.zeros((0, 0), np.uint16)
    gid = rows[:, 0].astype(np.int64) | (rows[:, 1].astype(np.int64) << 31)
    return [GameRecord(int(gid[i]), mv[i, :int(rows[i, 2])],
                       None if rows[i, 3] in (RESULT_NONE_WIRE, RESULT_TRUNCATED_WIRE) else int(rows[i, 3]),
                       bool(rows[i, 4]), truncated=bool(rows[i, 3] == RESULT_TRUNCATED_WIRE)) for i in range(len(rows))]


def gather_blocks(block, device=None, stats=None, force_collective=False):
    """Every rank's wire block on every rank: (rows of all ranks concatenated in rank order,
    per-rank row counts).  Two collectives -- (count, longest record) of every rank, then the blocks
    padded to the largest count and TRIMMED to the longest record of any rank (a block is packed
    for max_plies, 4.1 KB per row at 2048, while a game is ~320 plies: 0.65 KB) -- and ONE
    device-to-host copy each; RCCL over xGMI with the ``nccl`` backend, ``gloo`` in the CPU tests.  ``stats`` (a dict) receives the bytes moved and the wall time of the
    exchange (host staging included).  A world of one returns the block as it is unless
    ``force_collective`` (self-test of the RCCL path on a 1-GPU box: communicator, both
    all_gathers and the device-to-host copies run on a single rank)."""
    import time
    import torch
    import torch.distributed as dist
    block = np.ascontiguousarray(block, dtype=np.int32)
    grouped = dist.is_available() and dist.is_initialized()
    if not grouped or (dist.get_world_size() == 1 and not force_collective):
        if stats is not None:
            stats.update(world=1, rows=int(block.shape[0]), bytes_gathered=0, ms=0.0)
        return block, [block.shape[0]]
    world = dist.get_world_size()
    nccl = dist.get_backend() == "nccl"
    dev = device if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu"))
    t0 = time.perf_counter()
    need = HEADER + (int(block[:, 2].max()) + 1) // 2 if block.shape[0] else HEADER
    mine = torch.tensor([block.shape[0], min(need, block.shape[1])], dtype=torch.int64, device=dev)
    both = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(both, mine)
    both = both.cpu().view(world, 2)
    counts = [int(c) for c in both[:, 0].tolist()]
    nmax, w = max(counts), int(both[:, 1].max())
    padded = torch.zeros((nmax, w), dtype=torch.int32, device=dev)
    padded[:block.shape[0]] = torch.from_numpy(np.ascontiguousarray(block[:, :w])).to(dev)
    allb = torch.empty((world * nmax, w), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allb, padded)
    host = allb.cpu().numpy().reshape(world, nmax, w)
    if stats is not None:
        stats.update(world=world, rows=sum(counts), bytes_gathered=world * nmax * w * 4,
                     ms=(time.perf_counter() - t0) * 1e3, backend=dist.get_backend())
    return np.concatenate([host[r, :counts[r]] for r in range(world)], axis=0), counts


def gather_records(records, max_plies, device=None, stats=None):
    """All ranks' finished records on every rank, ordered by game id (identity without a
    process group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(records, key=lambda r: r.game_id)
    rows, _ = gather_blocks(pack(records, max_plies), device=device, stats=stats)
    return sorted(unpack(rows), key=lambda r: r.game_id)
