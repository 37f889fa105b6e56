"""CPU, world_size 2 over gloo: the record gather that follows the sharded self-play."""
import os

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd import records
    from chessrl_amd.game import uci_to_move
    # rank r owns games r, r+world, ...; rank 1 finished one more game than rank 0
    mine = [records.GameRecord(rank + world * k, [uci_to_move("e2e4")] * (3 + rank + k), k % 3 - 1,
                               bool((rank + k) & 1)) for k in range(2 + rank)]
    allr = records.gather_records(mine, max_plies=64)
    q.put((rank, [(r.game_id, len(r.moves), r.result, r.player_color) for r in allr]))
    dist.destroy_process_group()


def test_gather_records_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    assert [g[0] for g in got[0]] == [0, 1, 2, 3, 5]            # sorted by global game id
    assert got[0][1] == (1, 4, -1, True) and got[0][4] == (5, 6, 1, True)


def test_gather_records_single_process_is_identity():
    from chessrl_amd import records
    recs = [records.GameRecord(5, [1, 2], 0, True), records.GameRecord(1, [3], 1, False)]
    assert [r.game_id for r in records.gather_records(recs, 8)] == [1, 5]


def _bcast_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd.model import init_weights
    from chessrl_amd.train import broadcast_weights
    w = init_weights(1, 16, seed=100 + rank)                     # every rank starts different
    out = broadcast_weights(w, "cpu", src=0)
    ref = init_weights(1, 16, seed=100)
    q.put((rank, all(np.array_equal(out[k], ref[k]) for k in ref), sorted(out) == sorted(ref)))
    dist.destroy_process_group()


def test_trained_weights_broadcast_two_ranks():
    """After rank 0 trained, every rank holds rank 0's weights (one flat broadcast)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok and keys for _, ok, keys in got)


def _empty_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd import records
    from chessrl_amd.game import uci_to_move
    mine = [] if rank == 0 else [records.GameRecord(1, [uci_to_move("d2d4")] * 5, 0, False),
                                 records.GameRecord(3, [], None, True)]      # an empty, unfinished game
    allr = records.gather_records(mine, max_plies=16)
    q.put((rank, [(r.game_id, len(r.moves), r.result) for r in allr]))
    dist.destroy_process_group()


def test_gather_records_with_an_empty_rank_and_an_empty_game():
    """Ragged input: one rank finished nothing, another holds a zero-ply game with result None."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_empty_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1] == [(1, 5, 0), (3, 0, None)]
