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
                                 records.GameRecord(3, [], None, True),      # an empty, unfinished game
                                 records.GameRecord(5, [uci_to_move("g1f3")] * 15, None, False, truncated=True)]
    st = {}
    allr = records.gather_records(mine, max_plies=4096, stats=st)
    # the blocks travel trimmed to the longest record of any rank (15 plies = 8 ints), not to max_plies
    assert st["bytes_gathered"] == 2 * 3 * (records.HEADER + 8) * 4, st
    q.put((rank, [(r.game_id, len(r.moves), r.result) + ((r.truncated,) if r.truncated else ()) for r in allr]))
    dist.destroy_process_group()


def test_gather_records_with_an_empty_rank_and_an_empty_game():
    """Ragged input: one rank finished nothing, another holds a zero-ply game with result None and a game
    the runner cut off at max_plies (result None, truncated)."""
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
    assert got[0] == got[1] == [(1, 5, 0), (3, 0, None), (5, 15, None, True)]   # the truncated flag travels


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


def _async_train_worker(rank, world, port, q):
    """run_rolling with a background trainer on rank 0 (the product's BackgroundTrainer with a slow CPU
    train_fn standing in for the GPU step) and the news counter in the periodic all_reduce: nobody waits
    for the trainer."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd import records
    from chessrl_amd.selfplay import BackgroundTrainer, SelfPlayRunner
    from chessrl_amd.train import broadcast_weights
    N, R = 2 * world, 3
    run = SelfPlayRunner.__new__(SelfPlayRunner)
    run.rank, run.world, run.round_size, run.total_games = rank, world, N, N * R
    run._round_done, run.finished = {}, []
    mine = [g for g in range(N * R) if g % world == rank]
    finish_at = {g: 10 * (k + 1) for k, g in enumerate(mine)}      # one game every tenth move: a round per 20 moves
    state = {"move": 0, "sims": 0}
    last = max(finish_at.values())
    run.active = lambda: np.array([state["move"] < last])

    def play_move():
        state["move"] += 1
        state["sims"] += 100                                       # this rank's simulation counter
        time.sleep(0.02)
        for g, m in finish_at.items():
            if m == state["move"]:
                run.finished.append(records.GameRecord(g, [1, 2, 3], 0, True))
                run._round_done[g // N] = run._round_done.get(g // N, 0) + 1

    run.play_move = play_move
    weights = {"w": np.zeros(3, np.float32), "meta.blocks": np.array(1), "meta.filters": np.array(8)}
    trained_during = []

    def slow_train(w, recs):                                       # 0.5 s per round, adds 1 to the weights
        t0 = state["sims"]
        time.sleep(0.5)
        trained_during.append(state["sims"] - t0)                  # simulations THIS rank ran meanwhile
        return dict(w, w=w["w"] + 1), [{"loss": 0.0}]

    bg = BackgroundTrainer(weights, train_fn=slow_train) if rank == 0 else None
    model = {"w": weights, "loaded_at": []}

    def on_round(r, recs):
        allr = records.gather_records(recs, max_plies=8)
        assert [x.game_id for x in allr] == list(range(N * r, N * r + N))      # every rank's share of the round
        if bg is not None:
            bg.submit(r, allr)

    def on_news(k):
        w = bg.latest()[0] if bg is not None else model["w"]
        model["w"] = broadcast_weights(w, "cpu", src=0)
        model["loaded_at"].append((k, state["move"], state["sims"]))

    poll = lambda: bg.ready() if bg is not None else 0
    sims_at_submit = []
    done = run.run_rolling(R, on_round=on_round, sync_every=2, poll=poll, on_news=on_news)
    if bg is not None:
        bg.drain()
    run.sync_news(poll, on_news)
    if bg is not None:
        bg.close()
    q.put((rank, done, float(model["w"]["w"][0]), model["loaded_at"], trained_during, state["sims"]))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_keep_playing_while_rank_0_trains_in_the_background(world):
    """Two (and eight: the node the scaling curve is run on) gloo ranks, three rolling rounds, a trainer that
    takes 0.5 s per round on rank 0's thread: rank 1's simulation counter (and rank 0's own) advances while the
    trainer works, every weight set reaches every rank through the news word of the periodic all_reduce at the
    same sync index, and after the final drain all hold the weights of the last round."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + (os.getpid() % 2000) + world
    ps = [ctx.Process(target=_async_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=300) for _ in ps)}
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for rank in range(world):
        done, w, loaded_at, _, sims = got[rank]
        assert done == 3 and w == 3.0                              # three rounds trained, last weights everywhere
        assert [k for k, _, _ in loaded_at] == sorted(k for k, _, _ in loaded_at) and loaded_at[-1][0] == 3
        # same sync index on every rank for every weight set (the collective inside on_news cannot dead-lock)
        assert [(k, m) for k, m, _ in loaded_at] == [(k, m) for k, m, _ in got[0][2]]
    # rank 0 itself played on while its trainer thread worked ...
    assert all(d > 0 for d in got[0][3][:2]), got[0][3]
    # ... and rank 1 was never held: it was still simulating after the first weight set had arrived (the last
    # sets may arrive together at the final drain, after the last move)
    loads1 = got[1][2]
    assert len(loads1) >= 2 and loads1[0][2] < got[1][4], (loads1, got[1][4])


def _failing_trainer_worker(rank, world, port, q):
    """As _async_train_worker, but rank 0's trainer dies on its second round."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd import records
    from chessrl_amd.selfplay import BackgroundTrainer, SelfPlayRunner
    N, R = 2 * world, 50                                           # far more rounds than the run will see
    run = SelfPlayRunner.__new__(SelfPlayRunner)
    run.rank, run.world, run.round_size, run.total_games = rank, world, N, N * R
    run._round_done, run.finished = {}, []
    mine = [g for g in range(N * R) if g % world == rank]
    finish_at = {g: 4 * (k + 1) for k, g in enumerate(mine)}
    state = {"move": 0}
    last = max(finish_at.values())
    run.active = lambda: np.array([state["move"] < last])

    def play_move():
        state["move"] += 1
        time.sleep(0.01)
        for g, m in finish_at.items():
            if m == state["move"]:
                run.finished.append(records.GameRecord(g, [1, 2, 3], 0, True))
                run._round_done[g // N] = run._round_done.get(g // N, 0) + 1

    run.play_move = play_move

    def train(w, recs):
        if w["w"][0] >= 1:
            raise ValueError("bad round")
        return dict(w, w=w["w"] + 1), [{"loss": 0.0}]

    bg = BackgroundTrainer({"w": np.zeros(1, np.float32)}, train_fn=train) if rank == 0 else None
    t0 = time.time()
    try:
        run.run_rolling(R, on_round=lambda r, recs: bg.submit(r, recs) if bg is not None else None, sync_every=2,
                        poll=lambda: bg.ready() if bg is not None else 0, on_news=lambda k: None)
        q.put((rank, "finished", state["move"], time.time() - t0))
    except RuntimeError as e:
        q.put((rank, str(e) + " / " + str(e.__cause__), state["move"], time.time() - t0))
    dist.destroy_process_group()


def test_a_failed_background_trainer_ends_every_rank_at_the_same_sync_index():
    """rank 0's trainer thread dies while the other ranks keep playing -- they used to sit in the next
    all_reduce until the process group's timeout.  The periodic all_reduce carries a failure word: every rank
    raises at the same sync index, the failing one with its own exception as the cause."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 39700 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_failing_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=120) for _ in ps)}
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert "background trainer failed" in got[0][0] and "bad round" in got[0][0], got
    assert "another rank reported a failure" in got[1][0], got
    assert got[0][1] == got[1][1] and got[0][1] < 200, got           # the same move, long before the games run out
    assert max(got[0][2], got[1][2]) < 30


def test_background_trainer_trains_in_order_and_surfaces_its_errors():
    """BackgroundTrainer (one process): rounds are trained in submission order, each from the weights the
    previous one produced; ``ready`` counts finished weight sets; an exception in the trainer thread is raised
    at the next call instead of being lost with the thread."""
    import time
    import pytest
    from chessrl_amd.selfplay import BackgroundTrainer
    seen = []

    def train(w, recs):
        seen.append((float(w["w"][0]), list(recs)))
        if recs == ["boom"]:
            raise ValueError("bad round")
        time.sleep(0.05)
        return dict(w, w=w["w"] + len(recs)), [{"loss": 1.0}]

    bg = BackgroundTrainer({"w": np.zeros(1)}, train_fn=train)
    assert bg.ready() == 0
    bg.submit(0, ["a", "b"])
    bg.submit(1, ["c"])
    bg.drain()
    assert bg.ready() == 2 and float(bg.latest()[0]["w"][0]) == 3.0
    assert seen == [(0.0, ["a", "b"]), (2.0, ["c"])] and [r for r, _, _ in bg.latest()[1]] == [0, 1]
    bg.submit(2, ["boom"])
    with pytest.raises(RuntimeError, match="background trainer failed"):
        bg.drain()
    with pytest.raises(RuntimeError):
        bg.ready()


def test_rolling_rounds_need_a_share_for_every_rank_and_count_empty_shares_as_complete():
    """``round_size < world`` is refused (a rank without a share could never report a round complete); with
    ``total_games`` not a multiple of the round size the last, partial round still counts, and a rank whose
    share of it is empty reports it complete."""
    import pytest
    from chessrl_amd.selfplay import SelfPlayRunner
    run = SelfPlayRunner.__new__(SelfPlayRunner)
    run.rank, run.world, run.round_size, run.total_games = 3, 4, 2, 8
    run._round_done, run.finished = {}, []
    with pytest.raises(ValueError, match="smaller than the number of ranks"):
        run.run_rolling(1)
    run.rank, run.world, run.round_size, run.total_games = 1, 2, 4, 9          # rounds: 0-3, 4-7, 8 (rank 0 only)
    assert [run._round_share(r) for r in range(3)] == [2, 2, 0]
    run._round_done = {0: 2, 1: 2}
    assert run.rounds_complete() == 3                                           # the empty share does not block
    run._round_done = {0: 2, 1: 1}
    assert run.rounds_complete() == 1


def _dp_news_worker(rank, world, port, q):
    """every rank has a trainer of its own (data-parallel mode): a weight set is loaded when EVERY rank has it"""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd import records
    from chessrl_amd.selfplay import BackgroundTrainer, SelfPlayRunner
    N, R = 2 * world, 2
    run = SelfPlayRunner.__new__(SelfPlayRunner)
    run.rank, run.world, run.round_size, run.total_games = rank, world, N, N * R
    run._round_done, run.finished = {}, []
    mine = [g for g in range(N * R) if g % world == rank]
    finish_at = {g: 5 * (k + 1) for k, g in enumerate(mine)}
    state = {"move": 0}
    run.active = lambda: np.array([state["move"] < max(finish_at.values()) + 60])

    def play_move():
        state["move"] += 1
        time.sleep(0.01)
        for g, m in finish_at.items():
            if m == state["move"]:
                run.finished.append(records.GameRecord(g, [1, 2, 3], 0, True))
                run._round_done[g // N] = run._round_done.get(g // N, 0) + 1

    run.play_move = play_move
    ready_at = []

    def train(w, recs):                                   # rank 1's trainer is three times slower
        time.sleep(0.1 * (1 + 2 * rank))
        ready_at.append(state["move"])
        return dict(w, w=w["w"] + 1), [{"loss": 0.0}]

    bg = BackgroundTrainer({"w": np.zeros(1)}, train_fn=train, group="own")      # (any non-None group: sets are kept)
    loads = []
    done = run.run_rolling(R, on_round=lambda r, recs: bg.submit(r, recs), sync_every=2, poll=bg.ready,
                           on_news=lambda k: loads.append((k, state["move"], float(bg.take(k)["w"][0]))), news="min")
    bg.drain()
    run.sync_news(bg.ready, lambda k: loads.append((k, state["move"], float(bg.take(k)["w"][0]))), news="min")
    bg.close()
    q.put((rank, done, loads, ready_at))
    dist.destroy_process_group()


def test_data_parallel_news_waits_for_the_slowest_trainer():
    """news="min": a weight set is loaded, on every rank at the same sync index, only once EVERY rank's own
    trainer has finished it, and it is that set -- not whatever a faster rank has finished since."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 39500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_dp_news_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=180) for _ in ps)}
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (d0, l0, r0), (d1, l1, r1) = got[0], got[1]
    assert d0 == d1 == 2
    assert [(k, m) for k, m, _ in l0] == [(k, m) for k, m, _ in l1]             # same set at the same move
    assert all(w == float(k) for k, _, w in l0 + l1) and l0[-1][0] == 2          # set k holds k training rounds
    for (k, m, _) in l1:
        assert m >= r1[k - 1]                                                     # not before the slow rank had it


def _guard_follow_worker(rank, world, port, q):
    """run_rolling on two ranks whose evaluators start in an auto-kept f16; rank 1's run-time guard 'fires' at its
    third move (its evaluator goes to hybrid).  The periodic all_reduce carries the arithmetic: rank 0 follows at the
    next sync index through the evaluator's enter_strict."""
    import types
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chessrl_amd import records
    from chessrl_amd.selfplay import SelfPlayRunner

    class Eval(object):
        precision, entered = "f16", []

        def enter_strict(self, why):
            if self.precision != "f16":
                return False
            self.entered.append((state["move"], str(why)))
            self.precision = "hybrid"
            return True

    N, R = 4, 2
    run = SelfPlayRunner.__new__(SelfPlayRunner)
    run.rank, run.world, run.round_size, run.total_games = rank, world, N, N * R
    run._round_done, run.finished = {}, []
    ev = Eval()
    run.engine = types.SimpleNamespace(evaluator=ev, dev=torch.device("cpu"))
    mine = [g for g in range(N * R) if g % world == rank]
    finish_at = {g: 6 * (k + 1) for k, g in enumerate(mine)}
    state = {"move": 0}
    last = max(finish_at.values())
    run.active = lambda: np.array([state["move"] < last])

    def play_move():
        state["move"] += 1
        if rank == 1 and state["move"] == 3:
            ev.precision = "hybrid"                                # this rank's own guard fired
        for g, m in finish_at.items():
            if m == state["move"]:
                run.finished.append(records.GameRecord(g, [1, 2, 3], 0, True))
                run._round_done[g // N] = run._round_done.get(g // N, 0) + 1

    run.play_move = play_move
    done = run.run_rolling(R, sync_every=2)
    q.put((rank, done, ev.precision, list(ev.entered), getattr(run, "mode_follows", 0)))
    dist.destroy_process_group()


def test_a_rank_whose_precision_guard_fired_takes_the_other_ranks_with_it():
    """the guard flips precision per rank; ranks could end up in different arithmetics with nothing
    recording that.  The strictest mode of any rank travels in the periodic all_reduce; the others follow there."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 41300 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_guard_follow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=120) for _ in ps)}
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == 2
    assert got[0][1] == got[1][1] == "hybrid"
    assert got[1][2] == [] and got[1][3] == 0                       # rank 1 switched by itself
    (move, why), = got[0][2]
    assert move == 4 and "another rank" in why and got[0][3] == 1   # the first sync index (every 2 moves) after move 3
