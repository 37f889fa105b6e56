"""GPU parity: whole self-play games (lockstep runner, refill of finished slots, records,
drop-in Game/Agent/SelfPlayTree objects) vs the reference-shaped CPU oracle."""
import numpy as np
import pytest

from oracle import mcts_oracle
from oracle.chess_oracle import OracleGame
from oracle.fakenet import FakeNet

pytestmark = pytest.mark.gpu


def oracle_game(net, gid, seed, sims, noise, max_moves=None):
    from chessrl_amd.selfplay import game_color
    agent = mcts_oracle.OracleAgent(net)
    return mcts_oracle.play_game(agent, max_iters=sims, noise=noise, player_color=game_color(seed, gid),
                                 rng=np.random.default_rng([seed, gid]), max_moves=max_moves)


def test_full_games_with_refill_match_oracle():
    """5 complete games through 3 lockstep slots (so slots are refilled), Dirichlet noise ON
    with per-game streams; every record must equal the oracle's game move for move."""
    from chessrl_amd.selfplay import SelfPlayRunner
    net = FakeNet(seed=21, prior_shift=30)
    seed, sims = 5, 6
    run = SelfPlayRunner(net.to("cuda:0"), n_parallel=3, sims=sims, seed=seed, noise=True,
                         total_games=5, max_plies=2048)
    recs = sorted(run.run(), key=lambda r: r.game_id)
    assert [r.game_id for r in recs] == [0, 1, 2, 3, 4]
    for r in recs:
        g = oracle_game(net, r.game_id, seed, sims, True)
        h = g.get_history()
        assert r.get_history()["moves"] == h["moves"], r.game_id
        assert r.result == h["result"] and r.result is not None
        assert r.player_color == g.player_color
    cnt = run.engine.ctx.counters()
    assert cnt["sims"] == run.sims_run
    run.close()


def test_a_game_that_fills_its_record_is_handed_over_truncated_and_the_others_finish():
    """max_plies = 64: a game still running when its record cannot take another full move is ended by
    the runner -- result None, ``truncated`` -- instead of raising the device's capacity error; its slot is
    refilled, every other game finishes, and every record (cut off or complete) equals the oracle's game
    move for move as far as it goes."""
    from chessrl_amd import records
    from chessrl_amd.selfplay import SelfPlayRunner
    net = FakeNet(seed=21, prior_shift=30)
    seed, sims, cap = 5, 6, 64
    run = SelfPlayRunner(net.to("cuda:0"), n_parallel=3, sims=sims, seed=seed, noise=True,
                         total_games=8, max_plies=cap)
    recs = sorted(run.run(), key=lambda r: r.game_id)
    assert [r.game_id for r in recs] == list(range(8))
    cut = [r for r in recs if r.truncated]
    assert cut and len(cut) < len(recs) and run.truncated_games == len(cut)
    for r in recs:
        h = oracle_game(net, r.game_id, seed, sims, True).get_history()
        if r.truncated:
            assert r.result is None and cap - 3 <= len(r.moves) <= cap and len(h["moves"]) > len(r.moves)
            assert r.get_history()["moves"] == h["moves"][:len(r.moves)], r.game_id
        else:
            assert r.get_history()["moves"] == h["moves"] and r.result == h["result"] is not None, r.game_id
    run.engine.ctx.sync()                                   # no sticky device error
    run.close()
    back = records.unpack(records.pack(recs, cap))          # the flag survives the wire format
    assert [(b.truncated, b.result) for b in back] == [(r.truncated, r.result) for r in recs]


def test_complete_games_with_the_real_tower_do_not_depend_on_the_policy_format():
    """48 complete games (refill, compaction of the thinning batch, noise on) with the fused HIP
    tower: once with the heads writing only the legal moves' probabilities (CRL_POLICY_LEGAL, the
    default) and once through full 1968-vectors -- the same records, move for move."""
    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner

    class FullVectorsOnly(object):                       # the same evaluator without the legal-label entry point
        accepts_bitplanes = True

        def __init__(self, model):
            self.model = model
            self.forward_into = model.forward_into

        def __call__(self, planes):
            return self.model(planes)

    model = ChessModel(blocks=2, filters=64, seed=4)
    recs = []
    for ev in (model, FullVectorsOnly(model)):
        run = SelfPlayRunner(ev, n_parallel=16, sims=12, seed=9, noise=True, total_games=48, max_plies=2048)
        assert run.engine.legal_priors == (ev is model)
        recs.append({r.game_id: r for r in run.run()})
        run.close()
    a, b = recs
    assert sorted(a) == sorted(b) == list(range(48))
    for k in a:
        assert a[k] == b[k], k
    assert len({len(r.moves) for r in a.values()}) > 10          # games of many different lengths


def test_two_rank_sharding_plays_the_same_games():
    """game id -> rank id % world: two 'ranks' run one after the other on this GPU must
    produce exactly the games a single rank produces (streams keyed by global game id)."""
    from chessrl_amd.selfplay import SelfPlayRunner
    net = FakeNet(seed=8, prior_shift=29).to("cuda:0")
    kw = dict(sims=5, seed=2, noise=True, max_plies=2048)
    one = SelfPlayRunner(net, n_parallel=4, total_games=4, **kw)
    base = {r.game_id: r for r in one.run()}
    one.close()
    got = {}
    for rank in range(2):
        r = SelfPlayRunner(net, n_parallel=2, total_games=4, rank=rank, world=2, **kw)
        got.update({x.game_id: x for x in r.run()})
        r.close()
    assert sorted(got) == sorted(base) == [0, 1, 2, 3]
    for k in base:
        assert got[k] == base[k]


def test_dropin_objects_match_oracle():
    """Game / Agent / SelfPlayTree objects (the reference's API surface) on the HIP path."""
    from chessrl_amd.agent import Agent
    from chessrl_amd.game import Game
    from chessrl_amd import mctree, netencoder
    from oracle import encoder_oracle
    net = FakeNet(seed=4, prior_shift=29)
    agent = Agent(True, model=net.to("cuda:0"))
    oagent = mcts_oracle.OracleAgent(net)
    g, og = Game(), OracleGame()
    assert agent.move_encodings == oagent.move_encodings
    for u in ["e2e4", "c7c5", "g1f3", "d7d6", "f1b5"]:
        assert g.move(u) and og.move(u)
    assert not g.move("e1g1x") and not g.move("00000") and not g.move("a1a8")
    assert g.get_legal_moves() == og.get_legal_moves() and len(g) == len(og) == 5
    assert g.turn == og.turn and g.get_result() is None and g.get_fen() == og.get_fen()
    assert np.array_equal(netencoder.get_game_state(g), encoder_oracle.get_game_state(og))
    assert np.array_equal(netencoder.get_game_state(g, flipped=True),
                          encoder_oracle.get_game_state(og, flipped=True))
    c = g.get_copy()
    assert c.move("b8c6") and len(c) == 6 and len(g) == 5          # deep copy incl. move stack

    class _Move(object):                                           # what a python-chess Board looks like from outside
        def __init__(self, u):
            self.u = u

        def uci(self):
            return self.u

    class _Board(object):
        def __init__(self, fen, moves):
            self._fen, self.move_stack = fen, [_Move(u) for u in moves]

        def root(self):
            return _Board(self._fen, [])

        def fen(self):
            return self._fen

    start = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
    gb = Game(board=_Board(start, ["e2e4", "c7c5", "g1f3", "d7d6", "f1b5"]))     # game.py:17-21
    assert len(gb) == 5 and gb.get_fen() == g.get_fen() and gb.get_legal_moves() == g.get_legal_moves()
    assert np.array_equal(netencoder.get_game_state(gb), netencoder.get_game_state(g))   # history planes too
    with pytest.raises(ValueError):
        Game(board=_Board(start, ["e2e5"]))
    gb.free()
    assert c.get_history()["moves"][:5] == g.get_history()["moves"]
    # predict_* / best_move(real_game=True)
    assert np.array_equal(np.array(agent.predict_policy(g), dtype=np.float32),
                          np.array(oagent.predict_policy(og), dtype=np.float32))
    assert agent.predict_outcome(g) == oagent.predict_outcome(og)
    assert agent.best_move(g, real_game=True) == oagent.best_move(og, real_game=True)
    # SelfPlayTree.search_move
    tree = mctree.SelfPlayTree(g, threads=6)
    mv = tree.search_move(agent, max_iters=40, noise=False, ai_move=True)
    r = mcts_oracle.search(og, oagent, 40, noise=False)
    assert mv == r.moves
    assert [c.visits for c in tree.root.children] == r.visits and tree.root.visits == r.root_visits
    # the tree holds its own snapshot of the root (mctree.py:105-109 Node(root.get_copy())): child
    # states read AFTER the caller advanced its game are still root + our move + the reply
    plies_before = len(g)
    assert g.move(mv[0]) and g.move(mv[1]) and len(g) == plies_before + 2
    assert tree.root.state is not g and len(tree.root.state) == plies_before
    for ch in tree.root.children:
        hist = ch.state.get_history()["moves"]
        assert len(hist) == plies_before + (2 if ch.reply else 1)
        assert hist[plies_before] == ch.move and (ch.reply is None or hist[-1] == ch.reply)
    best = tree.root.children[int(np.argmax([c.visits for c in tree.root.children]))]
    assert best.state.get_fen() == g.get_fen()
    assert og.move(mv[0]) and og.move(mv[1])
    # agent.best_move(real_game=False) draws its noise from the global np.random stream
    np.random.seed(7)
    bm = agent.best_move(g, real_game=False, max_iters=40)
    np.random.seed(7)
    assert bm == oagent.best_move(og, real_game=False, max_iters=40, noise=True)
    g.free()
    c.free()


def test_dataset_roundtrip_and_reference_shaped_play_game(tmp_path):
    """DatasetGame (dataset.py) on HIP-backed Games + the reference-shaped play_game loop."""
    import random
    from chessrl_amd import selfplay
    from chessrl_amd.agent import Agent
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.game import Game
    net = FakeNet(seed=31, prior_shift=30)
    agent = Agent(True, model=net.to("cuda:0"))
    random.seed(4)
    np.random.seed(4)
    gam = selfplay.play_game(agent, max_iters=3)            # complete game, 3 sims/move
    random.seed(4)
    np.random.seed(4)
    og = mcts_oracle.play_game(mcts_oracle.OracleAgent(net), max_iters=3, noise=True)
    assert gam.get_history()["moves"] == og.get_history()["moves"]
    assert gam.get_result() == og.get_result() and gam.get_result() is not None
    d = DatasetGame()
    d.append(gam)
    text = str(d)
    d2 = DatasetGame()
    d2.loads(text)
    assert len(d2) == 1 and d2[0].get_history()["moves"] == gam.get_history()["moves"]
    assert d2[0].get_result() == gam.get_result()
    short = Game()                                          # augment a prefix: one Game copy per ply
    for m in gam.get_history()["moves"][:12]:
        assert short.move(m)
    aug = d.augment_game(short)
    assert len(aug) == 12 and aug[0]["game"].get_fen() == Game().get_fen()
    assert aug[3]["next_move"] == gam.get_history()["moves"][3] and len(aug[3]["game"]) == 3
    path = str(tmp_path / "gameplays.json")
    d.save(path)
    d.save(path)                                            # save() appends to what is on disk
    d3 = DatasetGame()
    d3.load(path)
    assert len(d3) == 2
    for g in aug:
        g["game"].free()


def test_dropin_play_game_replays_the_reference_games(golden_dir):
    """tests/golden/selfplay_games.json -- whole games by the reference's own selfplay.play_game ->
    AgentDistributed.best_move -> mctree.SelfPlayTree (noise on, seeded global streams) -- through the
    drop-in objects on the GPU: chessrl_amd.selfplay.play_game, Agent.best_move, SelfPlayTree, Game.
    Same colour, same moves, same result, 73 to 449 plies."""
    import json
    import os
    import random
    from chessrl_amd import selfplay
    from chessrl_amd.agent import Agent
    games = json.load(open(os.path.join(golden_dir, "selfplay_games.json")))["games"]
    assert len(games) >= 6
    for gm in games:
        agent = Agent(True, model=FakeNet(seed=gm["net_seed"], prior_shift=gm["prior_shift"]).to("cuda:0"))
        random.seed(gm["seed"])
        np.random.seed(gm["seed"])
        gam = selfplay.play_game(agent, max_iters=gm["sims"])
        h = gam.get_history()
        assert h["moves"] == gm["moves"], gm["seed"]
        assert gam.get_result() == gm["result"] and bool(h["player_color"]) == gm["player_color"]
        gam.free()


def test_dataset_matches_the_reference_dataset_module(golden_dir):
    """tests/golden/dataset_cases.json -- the reference's own dataset.DatasetGame (imported from
    /root/reference by oracle/make_golden.py) on two of the golden games: the JSON text it writes,
    loads() of that text, and augment_game's (position, next move, result) expansion -- against
    chessrl_amd.dataset.DatasetGame on HIP-backed Games."""
    import json
    import os
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.game import Game
    for c in json.load(open(os.path.join(golden_dir, "dataset_cases.json")))["cases"]:
        g = Game(player_color=c["player_color"], date=c["date"])
        for u in c["moves"]:
            assert g.move(u)
        d = DatasetGame()
        d.append(g)
        assert str(d) == c["json"]
        back = DatasetGame()
        back.loads(c["json"])
        assert len(back) == 1 and str(back) == c["json"]
        aug = d.augment_game(g)
        assert len(aug) == len(c["augment"])
        for a, e in zip(aug, c["augment"]):
            assert (len(a["game"]), a["game"].get_fen(), a["next_move"], a["result"]) == \
                (e["plies"], e["fen"], e["next_move"], e["result"])
            a["game"].free()
        back[0].free()
        g.free()


def test_training_batches_match_the_reference_data_sequence(golden_dir):
    """tests/golden/sequence_cases.json -- the reference's own netencoder.DataGameSequence (taken out of
    the parsed file and executed by oracle/make_golden.py over the reference's DatasetGame,
    get_game_state and label table) -- against chessrl_amd.netencoder.DataGameSequence, whose planes
    come from ONE launch of the HIP sequence-replay + encoder kernels: same 391 samples, same
    180-degree flips for the same np.random seed (neither game, either, both), same labels, results."""
    import hashlib
    import json
    import os
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.netencoder import DataGameSequence
    dsc = json.load(open(os.path.join(golden_dir, "dataset_cases.json")))["cases"]
    cases = json.load(open(os.path.join(golden_dir, "sequence_cases.json")))["cases"]
    assert len({c["x_sha256"] for c in cases}) == 4
    ds = DatasetGame()
    ds.loads("[" + ", ".join(c["json"][1:-1] for c in dsc) + "]")
    assert len(ds) == 2
    for c in cases:
        seq = DataGameSequence(ds, batch_size=2, random_flips=c["random_flips"])
        assert len(seq) == 1
        np.random.seed(c["seed"])
        x, (pol, val) = seq[0]
        assert x.shape == (c["n"], 8, 8, 127) and str(x.dtype) == c["x_dtype"] and pol.dtype == np.float32
        for k, i in enumerate((0, len(dsc[0]["moves"]))):
            ref = np.unpackbits(np.frombuffer(bytes.fromhex(c["first_sample_of_each_game_packbits_hex"][k]), np.uint8))
            assert np.array_equal(x[i].reshape(-1), ref[:8 * 8 * 127].astype(x.dtype)), (c["seed"], k)
        assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == c["x_sha256"], c["seed"]
        assert pol.shape == (c["n"], 1968) and (pol.sum(1) == 1).all()
        assert [int(i) for i in pol.argmax(1)] == c["labels"]
        assert [int(v) for v in val] == c["values"]
    for g in ds.games:
        g.free()


def test_cli_plays_and_trains_rounds(tmp_path):
    """``python -m chessrl_amd.selfplay modeldir --games N`` (selfplay.py:112-163): two rounds of
    play + train; records, weights and the training log land in modeldir."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "models")
    cmd = [sys.executable, "-m", "chessrl_amd.selfplay", d, "--games", "6", "--sims", "4",
           "--blocks", "1", "--filters", "64", "--rounds", "2", "--seed", "5"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = json.load(open(os.path.join(d, "gameplays.json")))
    assert len(recs) == 12 and all(g["result"] in (1, -1, 0) and len(g["moves"]) > 0 for g in recs)
    log = [json.loads(l) for l in open(os.path.join(d, "train_log.jsonl"))]
    assert len(log) == 2 and all(np.isfinite(e["loss"]) for e in log)
    w = dict(np.load(os.path.join(d, "model-0.npz")))
    assert int(w["meta.blocks"]) == 1 and int(w["meta.filters"]) == 64
    # a second invocation picks the saved model up (get_model_path) and only plays; the arithmetic
    # knobs of the CLI: the reference's pinned-numpy PUCT product and a pinned tower precision
    r = subprocess.run(cmd[:-4] + ["--no-train", "--numpy-promotion", "legacy", "--precision", "f16x3"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bad = subprocess.run(cmd[:-4] + ["--no-train", "--precision", "fp64"], cwd=root, capture_output=True, text=True)
    assert bad.returncode != 0 and "invalid choice" in bad.stderr
    assert np.array_equal(dict(np.load(os.path.join(d, "model-0.npz")))["stem.kernel"], w["stem.kernel"])


@pytest.mark.parametrize("agent_white", [False, True])
def test_game_agent_interactive_surface(agent_white):
    """GameAgent (gameagent.py:7-50): every human move is answered by the agent's greedy move."""
    from chessrl_amd.agent import Agent
    from chessrl_amd.gameagent import GameAgent
    net = FakeNet(seed=41, prior_shift=30)
    agent = Agent(agent_white, model=net.to("cuda:0"))
    oagent = mcts_oracle.OracleAgent(net)
    g = GameAgent(agent, player_color=not agent_white)
    og = OracleGame()
    rng = np.random.default_rng(12)
    if agent_white:
        assert g.move("a7a6") is True and len(g) == 1     # the argument is ignored: agent opens
        og.move(oagent.best_move(og, real_game=True))
    for _ in range(30):
        if og.get_result() is not None:
            break
        assert g.move("a1a1") is False and len(g) == len(og)   # illegal: ignored
        lm = og.get_legal_moves()
        mv = lm[int(rng.integers(len(lm)))]
        assert g.move(mv) is True
        og.move(mv)
        if og.get_result() is None:
            og.move(oagent.best_move(og, real_game=True))
        assert g.get_history()["moves"] == og.get_history()["moves"]
    c = g.get_copy()
    assert isinstance(c, GameAgent) and c.get_history()["moves"] == g.get_history()["moves"]
    with pytest.raises(ValueError):
        GameAgent(None)
    c.free()
    g.tearup()


def test_game_agent_replays_the_reference_game_agent(golden_dir):
    """tests/golden/game_agent_cases.json -- the reference's own GameAgent class (gameagent.py:7-50,
    executed by oracle/make_golden.py) driven by a scripted human: every ``move`` call's argument,
    return value and ply count afterwards (ignored first argument when the agent is white, refused
    illegal moves, the agent's greedy answers), the final move list and ``get_copy``."""
    import json
    import os
    from chessrl_amd.agent import Agent
    from chessrl_amd.gameagent import GameAgent
    for c in json.load(open(os.path.join(golden_dir, "game_agent_cases.json")))["cases"]:
        net = FakeNet(seed=c["net_seed"], prior_shift=c["prior_shift"])
        g = GameAgent(Agent(c["agent_white"], model=net.to("cuda:0")), player_color=not c["agent_white"])
        for call in c["calls"]:
            assert g.move(call["move"]) is call["returned"], call
            assert len(g) == call["plies"], call
        assert g.get_history()["moves"] == c["moves"] and g.get_result() == c["result"]
        cp = g.get_copy()
        assert isinstance(cp, GameAgent) is c["copy_is_game_agent"]
        assert cp.get_history()["moves"] == c["copy_moves"]
        cp.free()
        g.tearup()


def test_compaction_of_a_thinning_batch_keeps_every_game_identical():
    """A finite run stops refilling; the runner then halves the lockstep batch as games end
    (copying the running games into the first slots).  Records must not depend on that, and three
    of the games are checked against the oracle move for move."""
    from chessrl_amd.selfplay import SelfPlayRunner
    net = FakeNet(seed=17, prior_shift=30)
    kw = dict(n_parallel=256, sims=2, seed=9, noise=True, total_games=300, max_plies=2048)
    a = SelfPlayRunner(net.to("cuda:0"), compact=True, **kw)
    ra = {r.game_id: r for r in a.run()}
    assert a.G == 64 and a.engine.G == 64                   # 256 -> 128 -> 64 slots
    assert a.engine.ctx.counters()["sims"] == a.sims_run
    a.close()
    b = SelfPlayRunner(net.to("cuda:0"), compact=False, **kw)
    rb = {r.game_id: r for r in b.run()}
    assert b.G == 256
    b.close()
    assert sorted(ra) == sorted(rb) == list(range(300))
    for k in ra:
        assert ra[k] == rb[k], k
    longest = sorted(ra, key=lambda k: -len(ra[k]))[:3]     # these lived through every compaction
    for k in longest:
        g = oracle_game(net, k, 9, 2, True)
        assert ra[k].get_history()["moves"] == g.get_history()["moves"] and ra[k].result == g.get_result()


def test_cli_two_ranks_play_train_and_share_the_weights(tmp_path):
    """torchrun x 2 ranks (gloo, both on this GPU): games sharded by id, records gathered, rank 0
    trains, the weights are broadcast, the second round plays with them on both ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "models")
    env = dict(os.environ, CRL_DIST_BACKEND="gloo", CRL_DEVICE="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "chessrl_amd.selfplay", d,
           "--games", "6", "--sims", "4", "--blocks", "1", "--filters", "64", "--rounds", "2", "--seed", "5"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = json.load(open(os.path.join(d, "gameplays.json")))
    assert len(recs) == 12
    # round 0 is played with the same seeded random-init weights as the single-process run of
    # test_cli_plays_and_trains_rounds: sharding must not change those games
    one = str(tmp_path / "single")
    r1 = subprocess.run([sys.executable, "-m", "chessrl_amd.selfplay", one, "--games", "6", "--sims", "4",
                         "--blocks", "1", "--filters", "64", "--seed", "5", "--no-train"],
                        cwd=root, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    single = json.load(open(os.path.join(one, "gameplays.json")))
    assert [g["moves"] for g in recs[:6]] == [g["moves"] for g in single]
    assert len([l for l in open(os.path.join(d, "train_log.jsonl"))]) == 2


def test_rolling_rounds_play_the_same_games_and_hand_rounds_over_in_order():
    """Rolling rounds (SelfPlayRunner.run_rolling): the batch refills from the next round's ids while
    a round's long games finish.  What a game plays depends on its id only, so the three rounds of
    100 games must be, game for game, the games a one-shot run of 300 plays; rounds are handed
    over complete, in order, each as soon as its last game has ended (earlier than the end of the
    run for all but the last), and a weight update between rounds -- in place, under the captured
    hipGraph -- takes effect for the games still running."""
    from chessrl_amd.selfplay import SelfPlayRunner
    net = FakeNet(seed=17, prior_shift=30)
    kw = dict(n_parallel=128, sims=2, seed=9, noise=True, total_games=300, max_plies=2048)
    b = SelfPlayRunner(net.to("cuda:0"), **kw)
    rb = {r.game_id: r for r in b.run()}
    b.close()
    a = SelfPlayRunner(net.to("cuda:0"), round_size=100, **kw)
    handed, moves_at = [], []

    def on_round(r, recs):
        handed.append((r, sorted(x.game_id for x in recs)))
        moves_at.append(a.moves_played)
        for x in recs:
            assert x == rb[x.game_id], x.game_id

    assert a.run_rolling(3, on_round=on_round) == 3
    assert [r for r, _ in handed] == [0, 1, 2]
    assert [ids for _, ids in handed] == [list(range(100)), list(range(100, 200)), list(range(200, 300))]
    assert moves_at[0] < moves_at[1] < moves_at[2] and not a.finished and not a.active().any()
    a.close()
    # one round per 64 ids on a 2-rank shard: shares are 32 ids each and complete independently
    c = SelfPlayRunner(net.to("cuda:0"), n_parallel=32, sims=2, seed=9, noise=True, total_games=128,
                       max_plies=2048, round_size=64, rank=1, world=2)
    assert c._round_share(0) == 32 and c._round_share(1) == 32 and c._round_share(2) == 0
    while c.rounds_complete() < 2:
        c.play_move()
    assert sorted(x.game_id for x in c.take_round(0)) == list(range(1, 64, 2))
    assert all(x == rb[x.game_id] for x in c.take_round(1))
    c.close()


def test_cli_rolling_rounds_with_data_parallel_training_end_on_the_same_weights(tmp_path):
    """``--rolling --train-mode dp`` on two ranks (torchrun x 2, gloo, both on this GPU): every rank's background
    trainer trains on ITS games, the gradients are averaged over the trainer threads' own process group, one
    Adam step per two games; both ranks load every weight set at the same sync index and end on bit-identical
    weights without a broadcast."""
    import json
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "models")
    env = dict(os.environ, CRL_DIST_BACKEND="gloo", CRL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(29890 + os.getpid() % 40),
           "-m", "chessrl_amd.selfplay", d, "--games", "6", "--sims", "4", "--blocks", "1", "--filters", "64",
           "--rounds", "2", "--seed", "5", "--rolling", "--parallel", "4", "--train-mode", "dp"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = json.load(open(os.path.join(d, "gameplays.json")))
    assert len(recs) == 12 and all(g["result"] in (1, -1, 0) for g in recs)
    log = [json.loads(l) for l in open(os.path.join(d, "train_log.jsonl"))]
    assert len(log) == 2 and all(e["ranks"] == 2 and np.isfinite(e["loss"]) for e in log)
    ends = dict(re.findall(r"rank (\d): weights at the end ([0-9a-f]{16})", r.stderr))
    assert sorted(ends) == ["0", "1"] and ends["0"] == ends["1"], ends
    loaded = re.findall(r"rank (\d): weight set (\d) loaded", r.stderr)
    assert ("0", "2") in loaded and ("1", "2") in loaded


@pytest.mark.parametrize("ranks", [1, 2])
def test_cli_rolling_rounds_play_train_and_swap_weights_in_place(tmp_path, ranks):
    """``--rolling``: two overlapping rounds of 6 games through the CLI -- records of both rounds in id
    order per round, one training pass per round, the weights swapped in place under the running
    engine (and broadcast to the second rank: torchrun x 2, gloo, both ranks on this GPU)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "models")
    args = ["-m", "chessrl_amd.selfplay", d, "--games", "6", "--sims", "4", "--blocks", "1", "--filters", "64",
            "--rounds", "2", "--seed", "5", "--rolling", "--parallel", "4"]
    env = dict(os.environ)
    if ranks == 1:
        cmd = [sys.executable] + args
    else:
        env.update(CRL_DIST_BACKEND="gloo", CRL_DEVICE="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(29950 + os.getpid() % 40)] + args
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = json.load(open(os.path.join(d, "gameplays.json")))
    assert len(recs) == 12 and all(g["result"] in (1, -1, 0) and len(g["moves"]) > 0 for g in recs)
    log = [json.loads(l) for l in open(os.path.join(d, "train_log.jsonl"))]
    assert len(log) == 2 and all(np.isfinite(e["loss"]) for e in log)
    # round 0 is played on the initial weights until it is handed over: its first games (which end
    # before any training) are the games of a plain run with the same seed
    one = str(tmp_path / "plain")
    r1 = subprocess.run([sys.executable, "-m", "chessrl_amd.selfplay", one, "--games", "12", "--sims", "4",
                         "--blocks", "1", "--filters", "64", "--seed", "5", "--no-train", "--parallel", "4"],
                        cwd=root, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    plain = json.load(open(os.path.join(one, "gameplays.json")))
    assert [g["moves"] for g in recs[:6]] == [g["moves"] for g in plain[:6]]


def test_stepwise_driving_plays_the_same_games_as_whole_moves():
    """bench.py drives the runner one lockstep step at a time (``step()``: the Dirichlet noise of a
    move is drawn half-way through it, while the GPU works); ``run()`` drives whole moves.  Same games,
    and both equal the oracle's."""
    from chessrl_amd.selfplay import SelfPlayRunner
    net = FakeNet(seed=21, prior_shift=30)
    kw = dict(n_parallel=5, sims=6, seed=5, noise=True, total_games=9, max_plies=2048)
    a = SelfPlayRunner(net.to("cuda:0"), **kw)
    whole = {r.game_id: r for r in a.run()}
    a.close()
    b = SelfPlayRunner(net.to("cuda:0"), **kw)
    steps = 0
    while b.active().any():
        b.step()
        steps += 1
    stepwise = {r.game_id: r for r in b.finished}
    assert b.engine.ctx.counters()["sims"] == b.sims_run and steps % 6 == 0
    b.close()
    assert sorted(whole) == sorted(stepwise) == list(range(9))
    for k in whole:
        assert whole[k] == stepwise[k], k
    for k in (0, 4, 8):
        g = oracle_game(net, k, 5, 6, True)
        assert stepwise[k].get_history()["moves"] == g.get_history()["moves"] and stepwise[k].result == g.get_result()
