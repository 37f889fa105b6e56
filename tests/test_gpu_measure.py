"""GPU: the measurement plumbing of round 6 must not change what is measured -- stamps captured into the step's
hipGraph (crl_stamp), the steps-per-graph knob -- plus the advisor's findings of round 5 on the tower seam
(workspace size checked by the library, the reply margin's guards, the guard's stickiness)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tower_oracle

pytestmark = pytest.mark.gpu


def test_stamps_captured_into_the_step_graph_leave_the_search_unchanged_and_telescope():
    """The stamped build of a step (one-thread crl_stamp kernels between the phases and around every trunk launch,
    captured into the hipGraph with them) searches the same trees as the plain one, at any number of steps per graph
    launch; the stamps it leaves are complete, ordered, and add up to the step."""
    from chessrl_amd.engine import LockstepEngine, StampRing, summarise_stamps
    from chessrl_amd.model import ChessModel
    model = ChessModel(blocks=2, filters=64, seed=3, precision="hybrid")
    model.HYBRID_MIN_BOARDS = 0                                      # (a batch this small would run plain f16x3)
    sims = 40

    def trees(stamped, spg):
        eng = LockstepEngine(model, n_games=64, max_sims=sims, steps_per_graph=spg)
        assert eng.STEPS_PER_GRAPH == (spg or LockstepEngine.STEPS_PER_GRAPH)
        eng.reset()
        ring = None
        if stamped:
            ring = StampRing(4096, eng.dev)
            eng.set_stamps(ring)
            assert model.stamp_fn is not None
        eng.search_begin()
        eng.prepare_graphs(sims)
        if ring is not None:
            ring.clear()                                             # (the root evaluation and the capture's warm-up stamped too)
        eng.run_steps(sims)
        eng.ctx.sim_backup(eng.pri_s2.data_ptr(), eng.val_s2.data_ptr())
        rc = eng.ctx.root_children(["nchild", "visits", "values", "moves", "replies", "root_visits"])
        stamps = ring.read() if ring is not None else None
        if stamped:
            eng.set_stamps(None)
            assert model.stamp_fn is None and any(k[1] for k in eng._graphs)     # the stamped graphs stay cached ...
            eng.drop_stamped_graphs()
            assert not any(k[1] for k in eng._graphs)                            # ... until they are dropped
        eng.close()
        return rc, stamps

    plain, _ = trees(False, None)
    stamped, st = trees(True, 4)
    single, _ = trees(False, 1)
    for other in (stamped, single):
        for k in ("nchild", "visits", "moves", "replies", "root_visits"):
            assert np.array_equal(plain[k], other[k]), k
        assert np.array_equal(plain["values"].view(np.uint64), other["values"].view(np.uint64))
    assert (plain["root_visits"] == sims + 1).all()
    out = summarise_stamps(st)
    assert out["steps"] == sims
    assert sum(out["parts"].values()) == pytest.approx(out["ms_per_step"], rel=1e-9)
    # a hybrid step: STEP, SELECTED, f16 trunk pair, indexed pair, S1_DONE, REPLIED, f16x3 pair = 10 stamps, + one per graph
    assert out["stamps_per_step"] == pytest.approx(10 + 1 / 4 - 1 / sims, abs=0.05)
    assert set(out["trunk"]) == {"f16", "f16x3", "f16x3 indexed"}
    for kind in out["trunk"].values():
        assert kind["launches_per_step"] == 1 and 0 < kind["min_ms"] <= kind["launch_ms"] <= kind["max_ms"] < 5.0
    assert all(t1 >= t0 for (_, t0), (_, t1) in zip(st[:-1], st[1:]))           # the device clock never runs backwards
    assert 0.01 < out["ms_per_step"] < 10.0


def test_the_library_refuses_a_workspace_that_is_too_small_for_the_batch():
    """the layer-wise 256-filter split-precision trunk took a bare workspace pointer; a buffer sized for a
    smaller batch was silently overrun.  crl_trunk_forward_x / _indexed now take its size and fail with CRL_ERR_ARG."""
    from chessrl_amd import _lib
    from chessrl_amd.model import ChessModel
    m = ChessModel(blocks=1, filters=256, seed=2, precision="f16x3")
    L, vp = _lib.lib(), ctypes.c_void_p
    n = 8
    need = int(L.crl_trunk_workspace_bytes(256, n, _lib.TRUNK_BITPLANES | _lib.TRUNK_SPLIT))
    assert need == 2 * n * 64 * 1024
    planes = torch.zeros((n, 128), dtype=torch.int64, device="cuda")
    heads = torch.zeros((n, 192), dtype=torch.float32, device="cuda")
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = vp(torch.cuda.current_stream().cuda_stream)

    def call(nbytes):
        return L.crl_trunk_forward_x(st, 256, _lib.TRUNK_BITPLANES | _lib.TRUNK_SPLIT, vp(planes.data_ptr()),
                                     vp(m._wtiles3.data_ptr()), vp(m._wbias.data_ptr()), None, n, 1,
                                     vp(m._head_w.data_ptr()), vp(m._head_b.data_ptr()), vp(heads.data_ptr()),
                                     vp(ws.data_ptr()), nbytes)
    assert call(need) == 0
    assert call(need - 1) == -1 and call(0) == -1                    # CRL_ERR_ARG, nothing launched
    lst = torch.zeros(_lib.LIST_HEADER + n, dtype=torch.int32, device="cuda")
    lst[0] = 1
    for nbytes, rc in ((need, 0), (need // 2, -1)):
        assert L.crl_trunk_forward_indexed(st, 256, vp(planes.data_ptr()), vp(m._wtiles3.data_ptr()), vp(m._wbias.data_ptr()),
                                           n, 1, vp(m._head_w.data_ptr()), vp(m._head_b.data_ptr()), vp(heads.data_ptr()),
                                           vp(lst.data_ptr()), vp(ws.data_ptr()), nbytes) == rc
    torch.cuda.synchronize()
    # the model lends ONE buffer, sized for the largest batch it has seen, to every batch size
    a = m._trunk_workspace(64)
    epoch = m.graph_epoch
    assert m._trunk_workspace(8) is a and m._trunk_workspace(64) is a and m.graph_epoch == epoch
    big = m._trunk_workspace(max(128, 2 * a.numel() // (128 * 1024)))
    assert big.numel() > a.numel() and m.graph_epoch == epoch + 1    # grown: captured graphs hold the old address


def test_reply_margin_guards_zero_probabilities_a_cap_and_stickiness():
    """a policy entry that underflows to 0 in f16 made the log-distance infinite (every S1 board evaluated
    twice until the next weight set); the widened margin is capped; a run whose guard fired stays strict across
    reloads unless the probe passes with margin."""
    from chessrl_amd.model import ChessModel
    from tests.test_gpu_tower import _positions
    x, planes = _positions()
    assert ChessModel._log_distance(torch.tensor([0.0, 0.5]), torch.tensor([0.1, 0.5])) == 0.0
    assert ChessModel._log_distance(torch.tensor([0.0]), torch.tensor([0.1])) is None
    assert ChessModel._log_distance(torch.tensor([0.2]), torch.tensor([0.0])) is None
    sharp = tower_oracle.calibrated_weights(6, 64, planes[:512], seed=7)
    model = ChessModel(weights=sharp, precision="hybrid")
    probe = model._probe_margin
    assert probe == model.reply_margin > 0
    model._probe_margin = probe * 1e-6                               # as if the probe had seen nothing: the cap bites
    model.reply_margin = probe * 1e-6
    model._publish_reply_margin()
    d = model.margin_check(x[:256])
    assert d is not None and model.guard["margin_capped"] == 1 and model.guard["margin_widened"] == 1
    assert model.reply_margin == pytest.approx(model.MARGIN_CAP * probe * 1e-6)
    assert float(model._reply_margin_dev[0]) == pytest.approx(model.reply_margin, rel=1e-6)

    class Lenient(ChessModel):
        PROBE_TOL = 1.0
        STICKY_FACTOR = 1e-9
    m = Lenient(weights=sharp)
    assert m.precision == "f16" and not m.precision_probe["sticky_after_guard"]
    assert m.enter_strict("test") and m.precision == "hybrid" and m.guard["fired"]["why"] == "test"
    assert not m.enter_strict("again")
    epoch = m.graph_epoch
    m.load_dict(sharp)                                               # the next weight set of the run: stays strict
    assert m.precision == "hybrid" and m.precision_probe["sticky_after_guard"] and m.graph_epoch == epoch
    assert Lenient(weights=sharp).precision == "f16"                 # (another run starts afresh)


def test_the_reply_rule_of_hybrid_is_watched_on_the_runs_own_positions_and_heals_itself():
    """hybrid's reply rule (a board keeps its f16 reply unless its two best legal moves are closer than the margin) rests on
    a SAMPLED distance; the hand-over that re-measures the distance also checks the rule: every board whose reply differs
    between f16 and f16x3 must be one the margin lists.  With the product's margin no unlisted difference appears; with a
    margin forced to nothing the differing boards ARE unlisted: counted, and the margin is widened past them at once."""
    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    from tests.test_gpu_tower import _positions
    _, planes = _positions()
    sharp = tower_oracle.calibrated_weights(6, 64, planes[:512], seed=7)
    model = ChessModel(weights=sharp, precision="hybrid")
    model.HYBRID_MIN_BOARDS = 0
    run = SelfPlayRunner(model, n_parallel=512, sims=8, seed=3, noise=True, max_plies=512)
    run.GUARD_EVERY = 1
    for _ in range(6):
        run.play_move()
    rule = model.guard["reply_rule"]
    assert rule["checks"] == 6 and rule["boards"] > 2000 and rule["listed"] > 0
    assert rule["differ_but_not_listed"] == 0                          # the rule holds on this run's positions
    assert rule["replies_that_differ"] >= 0 and rule["largest_gap_of_a_differing_reply"] < model.reply_margin
    # a margin of nothing lists nobody: boards whose replies differ are now failures of the rule -- found and healed
    eng = run.engine
    seen = 0
    for _ in range(40):
        run.GUARD_EVERY = 0
        run.play_move()
        model.reply_margin = 1e-12
        model._publish_reply_margin()
        out = model.reply_rule_check(eng.planes_s2, eng._lab_s2[0], eng._lab_s2[1])
        assert out["listed"] <= out["differ"] + 4                    # (next to nothing is listed under that margin)
        if out["unlisted"]:
            seen += out["unlisted"]
            assert model.reply_margin >= 1.25 * 1e-12 and model.reply_margin > 1e-9       # widened past the failing board's gap
            break
    assert seen > 0 and model.guard["reply_rule"]["differ_but_not_listed"] == seen
    run.close()
