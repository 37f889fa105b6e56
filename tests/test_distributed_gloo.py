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


def _rolling_worker(rank, world, port, q):
    """SelfPlayRunner.run_rolling's control flow on a runner without a GPU: ``play_move`` is a script
    of which game ids finish at which move; everything else (round shares, take_round, the
    all_reduce(MIN) agreement, idling of a rank whose batch ran dry) is the product's code."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd import records
    from chessrl_amd.selfplay import SelfPlayRunner
    N, R = 8, 3                                                    # rounds of 8 global ids, 3 rounds
    run = SelfPlayRunner.__new__(SelfPlayRunner)
    run.rank, run.world, run.round_size, run.total_games = rank, world, N, N * R
    run._round_done, run.finished = {}, []
    mine = [g for g in range(N * R) if g % world == rank]
    # rank 0 finishes one game per move, rank 1 one game every third move: their shares of a round
    # complete at different moves, and rank 0's batch runs dry long before rank 1's
    finish_at = {g: (k + 1) * (1 if rank == 0 else 3) for k, g in enumerate(mine)}
    state = {"move": 0}
    run.active = lambda: np.array([state["move"] < max(finish_at.values())])

    def play_move():
        state["move"] += 1
        for g, m in finish_at.items():
            if m == state["move"]:
                run.finished.append(records.GameRecord(g, [1, 2, 3], 0, True))
                run._round_done[g // N] = run._round_done.get(g // N, 0) + 1

    run.play_move = play_move
    log = []

    def on_round(r, recs):
        gathered = records.gather_records(recs, max_plies=8)       # a collective inside the callback
        log.append((r, sorted(x.game_id for x in recs), [x.game_id for x in gathered]))

    done = run.run_rolling(R, on_round=on_round, sync_every=4)
    q.put((rank, done, log, run.finished))
    dist.destroy_process_group()


def test_rolling_rounds_agree_across_two_ranks():
    """Both ranks hand every round over in order, each with ITS share of the round's ids, inside the
    same on_round call (the record gather in it sees all 8 ids), and the loop ends on both although
    one rank's batch ran dry 36 moves before the other's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_rolling_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {r: (done, log, left) for r, done, log, left in (q.get(timeout=120) for _ in ps)}
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        done, log, left = got[rank]
        assert done == 3 and left == []
        assert [r for r, _, _ in log] == [0, 1, 2]
        for r, own, everyone in log:
            assert own == [g for g in range(8 * r, 8 * r + 8) if g % 2 == rank]
            assert everyone == list(range(8 * r, 8 * r + 8))


def test_host_helpers_of_round_3():
    """resolve_numpy_promotion follows the installed numpy; the launcher counts GPUs without HIP;
    noise rows drawn ahead give the choice the boundary draw gives."""
    import bench
    from chessrl_amd.engine import choose_children, dirichlet_row, resolve_numpy_promotion
    probe = "legacy" if (10 * np.float32(0.1)).dtype == np.float64 else "nep50"
    assert resolve_numpy_promotion("auto") == probe == ("nep50" if int(np.__version__.split(".")[0]) >= 2 else "legacy")
    assert resolve_numpy_promotion("legacy") == "legacy" and resolve_numpy_promotion("nep50") == "nep50"
    try:
        resolve_numpy_promotion("float128")
        raise AssertionError("accepted a bad mode")
    except ValueError:
        pass
    n = bench.visible_gpus()
    assert n is None or (isinstance(n, int) and n >= 0)
    os.environ["HIP_VISIBLE_DEVICES"] = "0"
    try:
        m = bench.visible_gpus()
        assert m is None or m <= 1
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]
    rng = np.random.default_rng(3)
    G = 6
    nchild = np.array([3, 0, 7, 1, 218, 20])
    visits = rng.integers(1, 50, (G, 256))
    root = np.array([int(visits[g, :nchild[g]].sum()) + 1 for g in range(G)])
    plies = np.array([0, 10, 29, 30, 77, 200])
    mk = lambda: [np.random.default_rng([5, g]) for g in range(G)]
    at_boundary = choose_children(visits, nchild, root, plies, noise=True, rngs=mk())
    ahead = [dirichlet_row(r, int(k)) if k else None for r, k in zip(mk(), nchild)]
    assert np.array_equal(choose_children(visits, nchild, root, plies, noise=True, rngs=None, noise_rows=ahead),
                          at_boundary)
    # the padded-matrix form the runner hands over (no per-game Python loop at the boundary)
    mat = np.zeros((G, 218))
    for g in range(G):
        if nchild[g]:
            mat[g, :nchild[g]] = ahead[g]
    assert np.array_equal(choose_children(visits, nchild, root, plies, noise=True, noise_rows=(mat, nchild.copy())),
                          at_boundary)
    wrong = nchild.copy()
    wrong[4] = 217
    try:
        choose_children(visits, nchild, root, plies, noise=True, noise_rows=(mat, wrong))
        raise AssertionError("accepted a noise matrix drawn for another child count")
    except RuntimeError:
        pass
    ahead[2] = ahead[2][:-1]                                        # a row of the wrong length is refused
    try:
        choose_children(visits, nchild, root, plies, noise=True, noise_rows=ahead)
        raise AssertionError("accepted a noise row of the wrong length")
    except RuntimeError:
        pass
