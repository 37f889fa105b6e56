"""GPU: the north_star's tower bar -- policy and value within 1e-3 of the fp32 reference network on
the same weights (model.py:31-63 runs fp32) -- on >= 4096 positions harvested from REAL self-play
(openings, middle games, long endgames, positions after promotions), for every tower size of
BASELINE.json plus the network the reference itself builds (10 x 256, model.py:33-37), and on two
kinds of weights:

* ``keras_default_init``: what bench.py times (random-init nets);
* ``sharp``: a net whose outputs are SENSITIVE to its input, the way a trained net's are
  (oracle/tower_oracle.calibrated_weights: BatchNorm statistics calibrated on real positions so
  every layer has unit gain, policy Dense scaled to peaked distributions, value spread over
  (-1, 1)).  The test asserts that sharpness (max p > 0.3, |v| > 0.5, values spread), so it cannot
  pass on a constant-output net: on such weights a near-uniform policy hides any error.

The fused trunk has two arithmetic modes: "f16" (one fp16 MFMA per product: BASELINE's "fp16 MFMA
inference", what bench.py times) and "f16x3" (operands carried as hi + lo fp16 pairs, three MFMAs
per product: fp32-grade).  Measured on MI355X (profiles/r03/tower_sharp_probe.json): f16 is 4e-4 ..
1.3e-3 off on default-init nets (a handful of the 4096 positions beyond 1e-3 at 10x128 and 10x256)
and 6e-3 .. 2.4e-2 off on the sharp ones; f16x3 is within 1e-4 on all.  ``precision="auto"`` (the
product's default) must therefore pick a mode that meets 1e-3 on the real positions -- asserted here
for what it picks, whatever that is -- and must pick f16x3 for the sharp nets; f16 must NOT meet the
bar there (negative control: the comparison is able to fail).  Plain f16 has ONE bar: 1e-3, required of every
net "auto" keeps in it -- on positions other than the probe's, for a grid of seeds -- and of no other.
"""
import numpy as np
import pytest
import torch

from oracle import tower_oracle
from tests.util import encode_prefixes, selfplay_position_prefixes

pytestmark = pytest.mark.gpu
N_POSITIONS = 4096
_CACHE = {}


def _positions():
    if "x" not in _CACHE:
        from chessrl_amd.model import ChessModel
        prefixes, info = selfplay_position_prefixes(N_POSITIONS)
        assert len(prefixes) == N_POSITIONS
        # real games: openings, long endgames and positions after promotions are all in the sample
        assert info["opening_lt_20"] > 50 and info["late_ge_150"] > 500 and info["after_a_promotion"] > 100
        x_bits, planes = encode_prefixes(ChessModel(blocks=2, filters=64, precision="f16"), prefixes)
        assert len(np.unique(planes.reshape(N_POSITIONS, -1), axis=0)) > 0.9 * N_POSITIONS
        _CACHE["x"], _CACHE["planes"], _CACHE["info"] = x_bits, planes, info
    return _CACHE["x"], _CACHE["planes"]


def _errors(pol, val, epol, eval_):
    dp = (pol.cpu() - epol).abs().max(1).values.numpy()
    dv = (val.cpu() - eval_).abs().numpy()
    return dp, dv


@pytest.mark.parametrize("weights", ["keras_default_init", "sharp"])
@pytest.mark.parametrize("blocks,filters", [(6, 64), (10, 128), (10, 256), (20, 256)])
def test_tower_within_1e3_on_real_selfplay_positions(blocks, filters, weights):
    from chessrl_amd.model import ChessModel
    x_bits, planes = _positions()
    if weights == "sharp":
        w = tower_oracle.calibrated_weights(blocks, filters, planes[:512], seed=7)
    else:
        w = tower_oracle.init_weights(blocks, filters, seed=4)
    epol, eval_ = tower_oracle.forward(w, planes)
    if weights == "sharp":
        # the net is sensitive: peaked policies, values spread over (-1, 1), different per position
        assert float(epol.max()) > 0.3 and float(epol.max(1).values.mean()) > 0.1
        assert float(eval_.abs().max()) > 0.5 and float(eval_.std()) > 0.3
    model = ChessModel(weights=w)                                   # precision="auto"
    assert model.fused and model.precision_probe["chosen"] == model.precision
    pol, val = model(x_bits)
    dp, dv = _errors(pol, val, epol, eval_)
    print("tower %dx%d %s: auto -> %s (probe %.2e / %.2e); %d real positions: |dpolicy| max %.2e p99.9 %.2e, "
          "|dvalue| max %.2e p99.9 %.2e" % (blocks, filters, weights, model.precision,
                                            model.precision_probe["dpolicy_max"], model.precision_probe["dvalue_max"],
                                            N_POSITIONS, dp.max(), np.quantile(dp, 0.999), dv.max(), np.quantile(dv, 0.999)))
    assert dp.max() <= 1e-3 and dv.max() <= 1e-3, (model.precision, dp.max(), dv.max())
    if weights == "sharp":
        assert model.precision == "hybrid"                          # every output under the bar in f16x3 arithmetic
        # negative control: one fp16 MFMA per product does NOT meet the bar on this net -- the
        # comparison above is able to fail; bounded drift (measured 6e-3 .. 2.4e-2)
        p16, v16 = model._forward_fused(x_bits, precision="f16")
        dp16, dv16 = _errors(p16, v16, epol, eval_)
        assert max(dp16.max(), dv16.max()) > 1e-3 and max(dp16.max(), dv16.max()) < 0.1
        # the split mode is fp32-grade, far inside the bar
        assert dp.max() <= 2e-4 and dv.max() <= 2e-4
    else:
        # the mode bench.py times (--precision f16) on these random-init weights
        p16, v16 = model._forward_fused(x_bits, precision="f16")
        dp16, dv16 = _errors(p16, v16, epol, eval_)
        print("    f16 on the same weights: |dpolicy| max %.2e, |dvalue| max %.2e p99.9 %.2e, %d of %d positions beyond 1e-3"
              % (dp16.max(), dv16.max(), np.quantile(dv16, 0.999), int(((dp16 > 1e-3) | (dv16 > 1e-3)).sum()), N_POSITIONS))
        # ONE bar for one mode: plain f16 is only REQUIRED inside 1e-3 on nets for which "auto" keeps
        # it -- that is the assertion above whenever model.precision == "f16", and
        # test_every_net_auto_keeps_in_f16_is_inside_1e3_on_positions_other_than_the_probe for a grid of seeds;
        # where "auto" left f16, f16 is merely reported (bounded drift: the comparison is sane)
        assert max(dp16.max(), dv16.max()) < 0.1
        if model.precision == "f16":
            assert dp16.max() <= 1e-3 and dv16.max() <= 1e-3


def test_every_net_auto_keeps_in_f16_is_inside_1e3_on_positions_other_than_the_probe():
    """VERDICT r4 (#4): the inverse of the tolerance.  For the grid of profiles/r04/auto_decisions_c3_seeds.json --
    eight random-init 10x128 nets as bench.py builds them (``--seed s``) plus two 6x64 ones -- every net that
    ``auto`` KEEPS in f16 must be inside 1e-3 of the fp32 oracle on the 4096 self-play positions of THIS file
    (other games, another net's play than the probe's).  If one fails, PROBE_TOL is too loose."""
    from chessrl_amd.model import ChessModel, init_weights
    x_bits, planes = _positions()
    kept, left = [], []
    for blocks, filters, seed in [(10, 128, s) for s in range(8)] + [(6, 64, 0), (6, 64, 2)]:
        w = init_weights(blocks, filters, seed)
        model = ChessModel(weights=w)
        probe = max(model.precision_probe["dpolicy_max"], model.precision_probe["dvalue_max"])
        if model.precision != "f16":
            left.append((blocks, filters, seed, probe))
            continue
        epol, eval_ = tower_oracle.forward(w, planes)
        pol, val = model(x_bits)
        dp, dv = _errors(pol, val, epol, eval_)
        kept.append((blocks, filters, seed, probe, float(dp.max()), float(dv.max())))
        print("%dx%d seed %d: probe %.2e -> f16 kept; %d other positions vs fp32: |dpolicy| %.2e |dvalue| %.2e"
              % (blocks, filters, seed, probe, N_POSITIONS, dp.max(), dv.max()))
    print("auto left f16 for:", [(b, f, s, "%.2e" % p) for b, f, s, p in left])
    assert kept, "the grid no longer holds a net auto keeps in f16: the test is vacuous"
    for b, f, s, probe, dp, dv in kept:
        assert probe <= ChessModel.PROBE_TOL and dp <= 1e-3 and dv <= 1e-3, (b, f, s, probe, dp, dv)


def test_the_run_time_guard_takes_an_auto_kept_f16_off_a_net_that_misses_the_bar_on_the_runs_own_positions():
    """the probe is 4096 positions of a fixed tiny net's games; a run evaluates millions of its own.
    ``SelfPlayRunner`` shows the model the tower inputs its search has just evaluated every GUARD_EVERY move
    boundaries (model.guard_check): a net that "auto" kept in f16 but that is beyond GUARD_TOL there leaves f16 for
    hybrid in mid-run.  Forced here: a sharp net with the probe's tolerance opened so that auto keeps f16."""
    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    _, planes = _positions()
    sharp = tower_oracle.calibrated_weights(6, 64, planes[:512], seed=7)

    class Lenient(ChessModel):
        PROBE_TOL = 1.0
    model = Lenient(weights=sharp)                                   # auto, and the probe lets everything pass
    assert model.precision == "f16" and model.guard["checks"] == 0
    run = SelfPlayRunner(model, n_parallel=64, sims=8, seed=3, noise=True, max_plies=512)
    run.GUARD_EVERY = 2
    epoch = model.graph_epoch
    for _ in range(2):
        run.play_move()
    assert model.guard["checks"] == 1 and model.guard["fired"] is not None and model.guard["worst"] > model.GUARD_TOL
    assert model.precision == "hybrid" and model.graph_epoch == epoch + 1
    run.play_move()                                                  # goes on in the compliant mode (graphs re-captured)
    assert model.guard["checks"] == 1                                # nothing left to guard
    run.close()
    # a mode asked for by name is not second-guessed; a soft net passes its checks
    named = ChessModel(weights=sharp, precision="f16")
    assert named.guard_check(_positions()[0][:64]) is None and named.precision == "f16"
    soft = ChessModel(blocks=2, filters=64, seed=1)
    d = soft.guard_check(_positions()[0][:256])
    assert soft.precision == "f16" and d is not None and d < soft.GUARD_TOL and soft.guard["positions"] == 256


def test_hybrid_reply_margin_follows_the_log_policy_distance_of_the_runs_own_positions():
    """hybrid's margin is HYBRID_K x a PROBE's largest |log p_f16 - log p_f16x3|.  The same
    hand-over as the f16 guard measures that distance on the run's own tree leaves and widens the margin where
    HYBRID_K x it is larger -- in the device float the captured graphs read: no re-capture, never narrower."""
    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    _, planes = _positions()
    sharp = tower_oracle.calibrated_weights(6, 64, planes[:512], seed=7)
    model = ChessModel(weights=sharp, precision="hybrid")
    model.HYBRID_MIN_BOARDS = 0                                      # (batches this small would run plain f16x3)
    probe_margin = model.reply_margin
    model.reply_margin = probe_margin * 1e-3                         # as if the probe had seen almost no difference
    model._publish_reply_margin()
    run = SelfPlayRunner(model, n_parallel=64, sims=8, seed=3, noise=True, max_plies=512)
    run.GUARD_EVERY = 2
    epoch = model.graph_epoch
    for _ in range(2):
        run.play_move()
    g = model.guard
    assert g["margin_checks"] == 1 and g["margin_widened"] == 1 and g["checks"] == 0 and g["fired"] is None
    assert model.reply_margin == model.HYBRID_K * g["worst_dlog"] > probe_margin * 1e-3
    assert abs(float(model._reply_margin_dev[0]) - model.reply_margin) <= 1e-6 * model.reply_margin
    assert model.graph_epoch == epoch and model.precision == "hybrid"
    wide = model.reply_margin * 1e6                                  # never narrower; and the captured steps read the
    model.reply_margin = wide                                        # float in place: under this margin every S1 board
    model._publish_reply_margin()                                    # is a close call
    model.margin_check(_positions()[0][:256])
    assert model.reply_margin == wide and model.guard["margin_widened"] == 1
    listed = model.fallback_boards()
    run.play_move()
    assert model.fallback_boards() >= listed + 64 and model.graph_epoch == epoch
    run.close()


def test_precision_modes_are_selectable_and_a_weight_swap_reselects():
    """``precision`` pins a mode; "auto" re-decides when new weights are loaded in place (training
    rounds) and says so through ``graph_epoch`` so that a LockstepEngine re-captures its hipGraph."""
    from chessrl_amd.engine import LockstepEngine
    from chessrl_amd.model import ChessModel
    _, planes = _positions()
    easy = tower_oracle.init_weights(2, 64, seed=4)
    sharp = tower_oracle.calibrated_weights(2, 64, planes[:256], seed=7)
    for mode in ("f16", "f16x3", "hybrid"):
        assert ChessModel(weights=sharp, precision=mode).precision == mode
    with pytest.raises(ValueError):
        ChessModel(weights=easy, precision="fp64")
    model = ChessModel(weights=easy)                                # 2x64 default init: f16 is far inside the bar
    assert model.precision == "f16" and model.graph_epoch == 0
    eng = LockstepEngine(model, n_games=8, max_sims=8)
    eng.reset()
    eng.search(8)
    first = dict(eng._graphs)
    assert first
    model.load_dict(sharp)                                          # in place, under the captured graph
    assert model.precision == "hybrid" and model.graph_epoch == 1 and model.reply_margin > 0
    eng.search(8)
    assert eng._graphs and all(eng._graphs.get(k) is not g for k, g in first.items())   # re-captured with the split kernels
    rc = eng.root_children()
    assert (rc["root_visits"] == 9).all()
    eng.close()


def test_reply_margin_lists_the_close_calls_and_the_indexed_trunk_evaluates_exactly_those():
    """The two entry points of the hybrid mode through the C-ABI: crl_reply_margin against numpy (both row
    formats: probabilities and logits; boards with fewer than two legal moves never listed), and
    crl_trunk_forward_indexed: listed rows of the head activations become the split-precision kernel's bits,
    every other row keeps the bits it had (64, 128 and 256 filters, lists that do not fill a workgroup)."""
    import ctypes
    from chessrl_amd import _lib, model as M
    from chessrl_amd.model import ChessModel
    vp = ctypes.c_void_p
    L = _lib.lib()
    st = vp(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(3)
    n = 300
    counts = rng.integers(0, 60, n).astype(np.int32)
    counts[:4] = [0, 1, 2, 218]
    for logits in (False, True):
        rows = rng.normal(0, 1.0, (n, 256)).astype(np.float32)
        if not logits:
            rows = np.exp(rows) / 50.0
        rows[5, :counts[5]] = rows[5, 0]                              # an exact tie is a close call
        thr = 0.05
        want = set()
        for b in range(n):
            if counts[b] >= 2:
                top = np.sort(rows[b, :counts[b]])[::-1][:2].astype(np.float64)
                margin = top[0] - top[1] if logits else np.log(top[0] / top[1])
                if margin < thr * (1 - 1e-4):
                    want.add(b)
                elif margin < thr * (1 + 1e-4):
                    want.add(-1)                                      # too close to the threshold to assert
        H = _lib.LIST_HEADER
        lst = torch.full((H + n,), -5, dtype=torch.int32, device="cuda")
        lst[2:4] = torch.tensor([2 ** 31 - 3, 0], dtype=torch.int32)  # the 64-bit running total, about to pass 2^31
        thr_dev = torch.tensor([thr], dtype=torch.float32, device="cuda")        # the margin is read from device memory
        assert L.crl_reply_margin(st, vp(torch.from_numpy(rows).cuda().data_ptr()), vp(torch.from_numpy(counts).cuda().data_ptr()),
                                  n, vp(thr_dev.data_ptr()), int(logits), vp(lst.data_ptr())) == 0
        got = lst.cpu().numpy()
        listed = set(got[H:H + got[0]].tolist())
        assert got[1] == -5 and int(got[2:4].view(np.int64)[0]) == 2 ** 31 - 3 + int(got[0]) and len(listed) == got[0]
        assert listed - want <= set() or -1 in want
        assert (want - {-1}) <= listed and 5 in listed and not ({0, 1} & listed)
    # (256 filters, 1024 boards: the layer-wise kernels' two indexed geometries -- one board per workgroup up to 256 listed
    # boards, two beyond -- on both sides of that boundary)
    for blocks, filters, n in ((1, 64, 64), (1, 64, 1024), (1, 128, 64), (1, 256, 64), (1, 256, 1024)):
        m = ChessModel(blocks=blocks, filters=filters, seed=5, precision="hybrid")
        planes = M._probe_bitplanes(m.device, 256)[:64].repeat(n // 64, 1).contiguous()
        planes[:, 3] ^= torch.arange(n, device="cuda")               # every board different
        _, h16 = m._run_fused(planes, precision="f16")
        _, h48 = m._run_fused(planes, precision="f16x3")
        assert not torch.equal(h16, h48)
        picks = [[0], [5, 17, 40], list(range(1, n, 3)), []]
        if n >= 512:
            picks += [list(range(2, 2 + 256)), list(range(3, 3 + 257)), list(range(n))]
        for pick in picks:
            lst = torch.zeros(_lib.LIST_HEADER + n, dtype=torch.int32, device="cuda")
            lst[0] = len(pick)
            lst[_lib.LIST_HEADER:_lib.LIST_HEADER + len(pick)] = torch.tensor(pick, dtype=torch.int32)[torch.randperm(len(pick))] if pick else 0
            hp = h16.clone()
            ws = m._trunk_workspace(n)                               # 256 filters: the layer-wise kernels' activation images
            assert L.crl_trunk_forward_indexed(st, filters, vp(planes.data_ptr()), vp(m._wtiles3.data_ptr()),
                                               vp(m._wbias.data_ptr()), n, blocks, vp(m._head_w.data_ptr()),
                                               vp(m._head_b.data_ptr()), vp(hp.data_ptr()), vp(lst.data_ptr()),
                                               vp(ws.data_ptr() if ws is not None else None), ws.numel() if ws is not None else 0) == 0
            torch.cuda.synchronize()
            mask = torch.zeros(n, dtype=torch.bool, device="cuda")
            mask[pick] = True
            assert torch.equal(hp[mask], h48[mask]) and torch.equal(hp[~mask], h16[~mask]), (filters, n, len(pick))


def _random_playouts(eng, target, picks):
    """the same random playouts for every engine: game g plays target[g] plies chosen by picks[ply][g]"""
    G = len(target)
    for ply in range(int(target.max())):
        moves, counts = eng.ctx.legal_moves()
        pick = (picks[ply] * np.maximum(counts, 1)).astype(np.int64)
        mv = np.where((target > ply) & (counts > 0), moves[np.arange(G), pick], 0xFFFF).astype(np.uint16)
        eng.ctx.push_moves(mv)


def test_a_weight_reload_under_a_captured_hybrid_graph_searches_with_the_new_margin():
    """the reply margin belongs to the weight set.  hybrid -> hybrid reloads keep the captured
    hipGraph (the weights are rewritten in place), so the margin must reach the kernel through device memory: a
    by-value argument would stay the margin of the weights the graph was captured with.  A graph captured on a
    soft net (margin ~1e-3) is replayed after ``load_dict`` of a sharp net (margin 30-100x larger): the trees, the
    replies and the number of boards evaluated twice are those of a fresh model of the sharp net driven WITHOUT
    graphs; with the stale margin the fall-back list would be a fraction of it."""
    from chessrl_amd.engine import LockstepEngine
    from chessrl_amd.model import ChessModel
    _, planes = _positions()
    soft = tower_oracle.init_weights(6, 64, seed=4)
    sharp = tower_oracle.calibrated_weights(6, 64, planes[:512], seed=7)
    G, sims = 512, 32
    rng = np.random.RandomState(12)
    target = rng.randint(0, 100, size=G)
    picks = rng.random_sample((int(target.max()), G))

    model = ChessModel(weights=soft, precision="hybrid")
    model.HYBRID_MIN_BOARDS = 0
    margin_soft = model.reply_margin
    eng = LockstepEngine(model, G, sims)
    eng.reset()
    _random_playouts(eng, target, picks)
    eng.search(sims)                                                 # captures the step graphs on the soft net
    graphs, epoch = dict(eng._graphs), model.graph_epoch
    assert graphs
    fb_soft = model.fallback_boards()
    model.load_dict(sharp)                                           # in place; hybrid stays hybrid: no re-capture
    assert model.precision == "hybrid" and model.graph_epoch == epoch
    assert model.reply_margin > 10 * margin_soft
    assert abs(float(model._reply_margin_dev.item()) - model.reply_margin) <= 1e-6 * model.reply_margin
    eng.reset()
    _random_playouts(eng, target, picks)
    eng.search(sims)
    assert all(eng._graphs.get(k) is g for k, g in graphs.items())   # the graphs captured on the soft net were replayed
    got, fb_got = eng.root_children(), model.fallback_boards() - fb_soft
    eng.close()

    fresh = ChessModel(weights=sharp, precision="hybrid")
    fresh.HYBRID_MIN_BOARDS = 0
    assert fresh.reply_margin == model.reply_margin
    ref = LockstepEngine(fresh, G, sims, use_graph=False)
    ref.reset()
    _random_playouts(ref, target, picks)
    ref.search(sims)
    want, fb_want = ref.root_children(), fresh.fallback_boards()
    ref.close()
    print("reload under a captured graph: margin %.2e -> %.2e; boards evaluated twice: soft %d, sharp %d (eager %d)"
          % (margin_soft, model.reply_margin, fb_soft, fb_got, fb_want))
    assert fb_got == fb_want and fb_want > 2 * max(fb_soft, 1)
    for k in ("nchild", "visits", "replies", "moves"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["values"].view(np.uint64), want["values"].view(np.uint64))
    assert np.array_equal(got["priors"].view(np.uint32), want["priors"].view(np.uint32))


def test_hybrid_mode_searches_the_trees_of_the_split_precision_mode():
    """precision="hybrid" on a sharp net: S2 in f16x3, S1's reply choice in f16 with the listed close calls
    evaluated again in f16x3.  512 games x 64 simulations from positions all over the game: the trees
    (visits, value sums, priors, stored replies) are those of the pure f16x3 search bit for bit -- for two
    consecutive moves under one captured hipGraph --, a minority of the S1 boards went through the fall-back,
    and the pure f16 search -- the negative control -- does NOT reproduce them (its replies differ somewhere)."""
    from chessrl_amd.engine import LockstepEngine
    from chessrl_amd.model import ChessModel
    _, planes = _positions()
    sharp = tower_oracle.calibrated_weights(6, 64, planes[:512], seed=7)
    G, sims = 512, 64
    rng = np.random.RandomState(11)
    target = rng.randint(0, 120, size=G)
    picks = rng.random_sample((int(target.max()), G))
    out = {}
    for mode in ("f16x3", "hybrid", "f16"):
        model = ChessModel(weights=sharp, precision=mode)
        model.HYBRID_MIN_BOARDS = 0                                  # (batches this small would run plain f16x3)
        eng = LockstepEngine(model, G, sims)
        eng.reset()
        _random_playouts(eng, target, picks)                         # the same random playouts for every mode
        eng.search(sims)
        first = eng.root_children()
        # a second move under the same captured graph (the list's counter must start from zero at every step:
        # a 4-byte memset node did not replay on this stack and the list overflowed from the second move on)
        chosen = np.where(first["nchild"] > 0, np.maximum(first["visits"].argmax(1), 0), -1).astype(np.int32)
        eng.advance(chosen)
        eng.search(sims)
        if mode == "hybrid":
            lst = model._fallback[G].cpu().numpy()
            assert 0 <= lst[0] <= G and len(set(lst[4:4 + lst[0]].tolist())) == lst[0]      # one step's list, no repeats
        out[mode] = (first, eng.ctx.counters(), model.fallback_boards() if mode == "hybrid" else 0, eng.root_children())
        eng.close()
    for k in (0, 3):
        a, b = out["f16x3"][k], out["hybrid"][k]
        assert np.array_equal(a["nchild"], b["nchild"]) and np.array_equal(a["visits"], b["visits"])
        assert np.array_equal(a["values"].view(np.uint64), b["values"].view(np.uint64))
        assert np.array_equal(a["priors"].view(np.uint32), b["priors"].view(np.uint32))
        assert np.array_equal(a["replies"], b["replies"]) and np.array_equal(a["moves"], b["moves"])
    a = out["f16x3"][0]
    s1 = out["hybrid"][1]["sims"]
    fb = out["hybrid"][2]
    print("hybrid: %d of %d S1 boards evaluated twice (%.1f %%)" % (fb, s1, 100.0 * fb / s1))
    assert 0 < fb < 0.3 * s1
    c = out["f16"][0]
    assert not (np.array_equal(a["replies"], c["replies"]) and np.array_equal(a["visits"], c["visits"]))


def test_hybrid_mode_plays_the_complete_games_of_the_split_precision_mode():
    """Whole self-play games on a sharp net (refill, compaction of the thinning batch, Dirichlet noise, several
    steps per hipGraph launch): the records played in ``hybrid`` are, move for move, the records played in pure
    ``f16x3`` -- ~60 000 S1 evaluations, every reply choice the same."""
    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    _, planes = _positions()
    sharp = tower_oracle.calibrated_weights(6, 64, planes[:512], seed=7)
    out = {}
    for mode in ("f16x3", "hybrid"):
        model = ChessModel(weights=sharp, precision=mode)
        model.HYBRID_MIN_BOARDS = 0
        run = SelfPlayRunner(model, n_parallel=32, sims=8, seed=13, noise=True, total_games=48, max_plies=1024)
        out[mode] = {r.game_id: r for r in run.run()}
        if mode == "hybrid":
            assert model.fallback_boards() > 0
        run.close()
    a, b = out["f16x3"], out["hybrid"]
    assert sorted(a) == sorted(b) == list(range(48))
    for k in a:
        assert a[k] == b[k], k
    assert len({len(r.moves) for r in a.values()}) > 10


@pytest.mark.parametrize("disturbance", ["disturb", "stream"])
def test_trunk_outputs_do_not_change_while_another_process_uses_the_gpu(disturbance):
    """Launch-to-launch identity of every trunk kernel family (64 / 128 / 256 filters, both
    precision modes, both workgroup geometries, and the indexed kernels of the hybrid mode) while a second
    process disturbs the GPU -- ``disturb``: small kernels and context churn: its workgroups share our CUs, LDS
    latencies jitter, and a kernel that reads a register before its hand-counted ``s_waitcnt lgkmcnt`` has
    made it valid shows different bits (static half: tools/check_asm_hazards.py); ``stream``: two 4-GiB
    buffers copied back and forth through every L2 channel: the weight tiles' LDS-DMA arrives late, and a
    ring slot read before its counted ``vmcnt`` wait + barrier shows (static half: tools/lds_race_check.py)."""
    import hashlib
    import os
    import subprocess
    import sys
    from chessrl_amd import model as M
    from chessrl_amd.model import ChessModel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import ctypes
    import time
    from chessrl_amd import _lib
    disturber = subprocess.Popen([sys.executable, os.path.join(root, "tools", "trunk_stability_probe.py"),
                                  disturbance, "100000"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(8)                                                    # let it get going (imports, allocation)
    vp = ctypes.c_void_p
    try:
        for blocks, filters, n in ((1, 64, 256), (1, 64, 2048), (2, 128, 256), (2, 128, 2048), (1, 256, 256), (1, 256, 1024)):
            m = ChessModel(blocks=blocks, filters=filters, seed=5, precision="f16")
            planes = M._probe_bitplanes(m.device, 256).repeat(n // 256, 1).contiguous()
            m.precision_requested = "auto"
            m._pack_fused(m.weights)
            for mode in ("f16", "f16x3"):
                seen = set()
                for _ in range(60):
                    _, hp = m._run_fused(planes, precision=mode)
                    torch.cuda.synchronize()
                    seen.add(hashlib.md5(hp.cpu().numpy().tobytes()).hexdigest())
                assert disturber.poll() is None, "the disturber ended early"
                assert len(seen) == 1, (blocks, filters, n, mode, len(seen))
            # the indexed split kernels (hybrid mode): every third board listed, onto f16 activations
            lst = torch.zeros(_lib.LIST_HEADER + n, dtype=torch.int32, device="cuda")
            pick = torch.arange(0, n, 3, dtype=torch.int32)
            lst[0] = len(pick)
            lst[_lib.LIST_HEADER:_lib.LIST_HEADER + len(pick)] = pick
            _, base = m._run_fused(planes, precision="f16")
            seen = set()
            for _ in range(40):
                hp = base.clone()
                assert _lib.lib().crl_trunk_forward_indexed(
                    vp(torch.cuda.current_stream().cuda_stream), filters, vp(planes.data_ptr()), vp(m._wtiles3.data_ptr()),
                    vp(m._wbias.data_ptr()), n, blocks, vp(m._head_w.data_ptr()), vp(m._head_b.data_ptr()),
                    vp(hp.data_ptr()), vp(lst.data_ptr()),
                    vp(m._trunk_workspace(n).data_ptr() if m._trunk_workspace(n) is not None else None),
                    m._trunk_workspace(n).numel() if m._trunk_workspace(n) is not None else 0) == 0
                torch.cuda.synchronize()
                seen.add(hashlib.md5(hp.cpu().numpy().tobytes()).hexdigest())
            assert len(seen) == 1, (blocks, filters, n, "indexed", len(seen))
    finally:
        disturber.kill()
        disturber.wait()
