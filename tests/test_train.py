"""Training step: chessrl_amd/train.py vs oracle/train_oracle.py.

CPU tests: the oracle's own known answers (Adam, BatchNorm moving statistics, loss terms) and the
product's host logic (TrainTower on CPU tensors: weight-dict round trip, loss / gradient / update
equality with the independently written oracle).  GPU tests: the same comparison on the device
through ``Trainer`` and ``DataGameSequence`` (encoder kernel), the reference-shaped
``Agent.train`` on a self-played game, and that a hipGraph-captured engine sees the new weights.

Tolerances (fp32 on both sides, different summation orders): loss 1e-5 relative (1e-4 absolute on
its policy / value terms); gradients 5e-3
of the tensor's max |g| (GPU GEMMs vs CPU; batch-statistics BN backward is ill-conditioned); BN moving statistics 1e-5; one Adam step from
IDENTICAL gradients 1e-7.
"""
import os

import numpy as np
import pytest
import torch

from oracle import encoder_oracle, tower_oracle, train_oracle
from oracle.chess_oracle import OracleGame, move_to_uci

BLOCKS, FILTERS = 2, 32


def _random_game(seed, plies):
    rng = np.random.default_rng(seed)
    g = OracleGame()
    while len(g) < plies and g.get_result() is None:
        lm = g.legal_move_ids()
        g.move(move_to_uci(lm[int(rng.integers(len(lm)))]))
    return g


def _oracle_batch(game, labels, res=1):
    """Augmented samples of an oracle game (dataset.py:21-43): planes, move indices, results.
    The random games are unfinished: ``res`` stands in for the final result."""
    moves = game.get_history()["moves"]
    g = OracleGame()
    planes, idx = [], []
    for m in moves:
        planes.append(encoder_oracle.get_game_state(g))
        idx.append(labels[m])
        g.move(m)
    return np.stack(planes), np.array(idx), np.full(len(moves), float(res), np.float32)


@pytest.fixture(scope="module")
def labels(golden_dir):
    import json
    import os
    lab = json.load(open(os.path.join(golden_dir, "uci_labels.json")))
    lab = lab["labels"] if isinstance(lab, dict) else lab
    return {u: i for i, u in enumerate(lab)}


# ------------------------------------------------------------------------------------------ CPU

def test_oracle_adam_known_answer():
    """Keras Adam by hand: g = [1, -2], step 1: m = 0.1 g, v = 0.001 g^2,
    lr_1 = 0.002 * sqrt(0.001) / 0.1, w -= lr_1 * m / (sqrt(v) + 1e-7)."""
    w = {"a.kernel": np.array([0.5, 0.5], np.float32), "meta.blocks": np.array(0)}
    st = train_oracle.AdamState(w)
    out = train_oracle.adam_update(w, {"a.kernel": np.array([1.0, -2.0], np.float32)}, st)
    lr1 = 0.002 * np.sqrt(0.001) / 0.1
    exp = [0.5 - lr1 * 0.1 / (np.sqrt(0.001) + 1e-7), 0.5 + lr1 * 0.2 / (np.sqrt(0.004) + 1e-7)]
    assert np.allclose(out["a.kernel"], exp, rtol=0, atol=1e-7)
    # the un-corrected epsilon: the first step is lr * g / (|g| + eps * sqrt(1 - b2)^-1 ...) ~ lr
    assert abs((0.5 - out["a.kernel"][0]) - 0.002) < 1e-6
    out2 = train_oracle.adam_update(out, {"a.kernel": np.array([1.0, -2.0], np.float32)}, st)
    assert st.t == 2 and abs((out["a.kernel"][0] - out2["a.kernel"][0]) - 0.002) < 1e-6


def test_oracle_loss_terms_and_bn_moving_stats(labels):
    w = tower_oracle.init_weights(BLOCKS, FILTERS, seed=2)
    planes, idx, res = _oracle_batch(_random_game(5, 24), labels)
    losses, grads, stats, p, v = train_oracle.loss_and_grads(w, planes, idx, res)
    # at Glorot init the policy is near uniform: crossentropy ~ log(1968)
    assert abs(losses["policy_out_loss"] - np.log(1968)) < 0.2
    assert abs(losses["value_out_loss"] - float(np.mean((res - v.numpy()) ** 2))) < 1e-6
    reg = sum(0.01 * float((np.asarray(w[k], np.float64) ** 2).sum()) for k in w if k.endswith(".kernel"))
    assert abs(losses["reg_loss"] - reg) < 1e-3 * reg
    assert abs(losses["loss"] - (losses["policy_out_loss"] + losses["value_out_loss"] + losses["reg_loss"])) < 1e-5
    # moving statistics: 0.99 * old + 0.01 * batch; a conv bias feeding a BN has zero gradient
    assert set(stats) == {k for k in w if k.endswith((".mean", ".var"))}
    assert np.all(np.abs(stats["policy.bn.var"] - 0.99) < 0.05)
    assert np.abs(grads["block0.conv1.bias"]).max() < 1e-6
    assert np.abs(grads["stem.kernel"]).max() > 1e-5
    # same forward in inference mode differs (moving stats are identity at init)
    p_inf, _ = tower_oracle.forward(w, planes)
    assert (p - p_inf).abs().max() > 0


def test_oracle_training_reduces_the_loss(labels):
    w = tower_oracle.init_weights(1, 16, seed=3)
    planes, idx, res = _oracle_batch(_random_game(6, 16), labels)
    st = train_oracle.AdamState(w)
    first = None
    for _ in range(8):
        w, losses, _ = train_oracle.train_step(w, planes, idx, res, st)
        first = first or losses["loss"]
    assert losses["loss"] < first - 0.5


def test_train_tower_matches_oracle_on_cpu(labels):
    """Host logic of the product (no GPU): TrainTower + keras_losses + KerasAdam on CPU tensors
    against the independently written oracle, two consecutive steps."""
    from chessrl_amd.train import KerasAdam, TrainTower, keras_losses
    w = tower_oracle.init_weights(BLOCKS, FILTERS, seed=4, randomize_bn=True)
    planes, idx, res = _oracle_batch(_random_game(7, 30), labels)
    net = TrainTower(BLOCKS, FILTERS)
    net.load_keras_dict(w)
    back = net.to_keras_dict()
    assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)
    net.train()
    opt = KerasAdam(net.parameters())
    x = torch.zeros((len(idx), 8, 8, 128))
    x[..., :127] = torch.from_numpy(planes).float()
    st = train_oracle.AdamState(w)
    ow = w
    for step in range(2):
        opt.zero_grad()
        p, v = net(x)
        total, cce, mse, reg = keras_losses(p, v, torch.from_numpy(idx), torch.from_numpy(res), net.regularized())
        total.backward()
        ow, ol, og = train_oracle.train_step(ow, planes, idx, res, st)
        assert abs(total.item() - ol["loss"]) <= 1e-5 * abs(ol["loss"])
        assert abs(cce.item() - ol["policy_out_loss"]) <= 1e-5 and abs(mse.item() - ol["value_out_loss"]) <= 1e-5
        g = net.stem.weight.grad[:, :127].permute(2, 3, 1, 0).numpy()
        assert np.abs(g - og["stem.kernel"]).max() <= 1e-4 * np.abs(og["stem.kernel"]).max()
        g = net.policy_fc.weight.grad.t().numpy()
        assert np.abs(g - og["policy.dense.kernel"]).max() <= 1e-4 * np.abs(og["policy.dense.kernel"]).max()
        opt.step()
        got = net.to_keras_dict()
        for k in ("block1.bn2.mean", "block1.bn2.var", "value.bn.var", "policy.bn.mean"):
            assert np.abs(got[k] - ow[k]).max() <= 1e-5, k
        # weights after the step: Adam turns gradient noise into +-lr on entries whose gradient is
        # ~0, so compare where the gradient is well above the noise floor
        for k in ("stem.kernel", "block0.conv2.kernel", "policy.dense.kernel", "value.dense1.kernel",
                  "block1.bn1.gamma"):
            mask = np.abs(og[k]) > 1e-3 * np.abs(og[k]).max()
            assert mask.sum() > 0 and np.abs(got[k] - ow[k])[mask].max() <= 5e-5, k


def test_keras_adam_equals_oracle_from_identical_gradients():
    from chessrl_amd.train import KerasAdam
    rng = np.random.default_rng(0)
    w = {"a.kernel": rng.normal(size=(7, 5)).astype(np.float32), "b.bias": rng.normal(size=9).astype(np.float32),
         "meta.blocks": np.array(0)}
    params = [torch.nn.Parameter(torch.from_numpy(w["a.kernel"].copy())),
              torch.nn.Parameter(torch.from_numpy(w["b.bias"].copy()))]
    opt = KerasAdam(params)
    st = train_oracle.AdamState(w)
    for step in range(5):
        grads = {"a.kernel": (rng.normal(size=(7, 5)) * 10.0 ** rng.integers(-6, 2)).astype(np.float32),
                 "b.bias": rng.normal(size=9).astype(np.float32)}
        params[0].grad = torch.from_numpy(grads["a.kernel"].copy())
        params[1].grad = torch.from_numpy(grads["b.bias"].copy())
        opt.step()
        w = train_oracle.adam_update(w, grads, st)
        assert np.abs(params[0].detach().numpy() - w["a.kernel"]).max() <= 1e-7
        assert np.abs(params[1].detach().numpy() - w["b.bias"]).max() <= 1e-7


def test_trainer_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chessrl_amd.train import Trainer
    with pytest.raises(RuntimeError):
        Trainer(tower_oracle.init_weights(1, 16), "cuda:0")


# ------------------------------------------------------------------------------------------ GPU

def _record(oracle_game, res=1, gid=0):
    """The slot-free record form the lockstep runner produces (records.GameRecord)."""
    from chessrl_amd.game import uci_to_move
    from chessrl_amd.records import GameRecord
    return GameRecord(gid, [uci_to_move(m) for m in oracle_game.get_history()["moves"]], res, True)


@pytest.mark.gpu
def test_data_sequence_matches_oracle_samples(labels):
    """DataGameSequence: planes from the encoder kernel, labels and results, 180-degree flip."""
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.netencoder import DataGameSequence
    og = [_random_game(11, 40), _random_game(12, 25)]
    ds = DatasetGame([_record(og[0], 1), _record(og[1], -1)])
    seq = DataGameSequence(ds, batch_size=2)
    assert len(seq) == 1
    x, (yp, yv) = seq[0]
    ex = [_oracle_batch(og[0], labels, 1), _oracle_batch(og[1], labels, -1)]
    assert x.shape == (65, 8, 8, 127) and x.dtype == np.float64
    assert np.array_equal(x, np.concatenate([e[0] for e in ex]))
    assert np.array_equal(yp.argmax(1), np.concatenate([e[1] for e in ex])) and (yp.sum(1) == 1).all()
    assert np.array_equal(yv, np.concatenate([e[2] for e in ex]))
    flipped = DataGameSequence(ds, batch_size=2, random_flips=1.0)
    xf, _ = flipped[0]
    assert np.array_equal(xf, np.stack([np.rot90(a, k=2) for a in x]))
    one = DataGameSequence(ds, batch_size=1)                  # agent.py:85: batch_size = 1 game
    assert len(one) == 2 and one[1][0].shape[0] == 25


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,filters", [(2, 32), (6, 64)])
def test_gpu_train_steps_match_oracle(labels, blocks, filters):
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.netencoder import DataGameSequence
    from chessrl_amd.train import Trainer
    w = tower_oracle.init_weights(blocks, filters, seed=8, randomize_bn=True)
    og = _random_game(13, 48)
    planes, idx, res = _oracle_batch(og, labels)
    seq = DataGameSequence(DatasetGame([_record(og)]), batch_size=1)
    x, y, z = seq.device_batch(0)
    tr = Trainer(w, "cuda:0")
    st = train_oracle.AdamState(w)
    ow = w
    checks = []                                               # (what, measured, tolerance)
    for step in range(3):
        logs = tr.train_on_batch(x, y, z)
        grads = {"stem.kernel": tr.net.stem.weight.grad[:, :127].permute(2, 3, 1, 0).cpu().numpy(),
                 "block1.conv2.kernel": tr.net.conv2[1].weight.grad.permute(2, 3, 1, 0).cpu().numpy(),
                 "policy.dense.kernel": tr.net.policy_fc.weight.grad.t().cpu().numpy(),
                 "value.dense2.kernel": tr.net.value_fc2.weight.grad.t().cpu().numpy(),
                 "block0.bn1.gamma": tr.net.bn1[0].weight.grad.cpu().numpy()}
        ow, ol, og_ = train_oracle.train_step(ow, planes, idx, res, st)
        checks.append(("step%d loss" % step, abs(logs["loss"] - ol["loss"]) / abs(ol["loss"]), 1e-5))
        checks.append(("step%d policy loss" % step, abs(logs["policy_out_loss"] - ol["policy_out_loss"]), 1e-4))
        checks.append(("step%d value loss" % step, abs(logs["value_out_loss"] - ol["value_out_loss"]), 1e-4))
        # trunk gradients pass through batch-statistics BN backward (differences of large sums):
        # measured 1e-6 .. 2.2e-3 of the tensor's max between MI355X and CPU fp32 (6 blocks, 48 samples)
        for k, g in grads.items():
            checks.append(("step%d grad %s" % (step, k), np.abs(g - og_[k]).max() / np.abs(og_[k]).max(), 5e-3))
        got = tr.weights()
        worst = max(np.abs(got[k] - ow[k]).max() / max(1.0, np.abs(ow[k]).max())
                    for k in got if k.endswith((".mean", ".var")))
        checks.append(("step%d BN moving statistics" % step, worst, 1e-5))
        # updated weights where the gradient is far above its noise: Adam maps gradient noise on
        # near-zero entries to +-lr, which is also why the oracle continues from the DEVICE's
        # weights below (its own Adam moments are kept) -- otherwise step 1 would compare two
        # trajectories that already differ by O(lr) on those entries (measured: 8 % on gradients)
        for k in ("stem.kernel", "policy.dense.kernel", "value.dense1.kernel"):
            mask = np.abs(og_[k]) > 1e-1 * np.abs(og_[k]).max()
            checks.append(("step%d updated %s" % (step, k), np.abs(got[k] - ow[k])[mask].max(), 2e-4))
        ow = got
    for what, v, tol in checks:
        print("%-40s %.3g (tol %.0e)" % (what, v, tol))
    bad = [c for c in checks if not c[1] <= c[2]]
    assert not bad, bad


@pytest.mark.gpu
def test_agent_train_on_a_self_played_game_updates_the_search_path(tmp_path):
    """selfplay.py:98-108 shape: play -> DatasetGame -> Agent.train -> save -> load; the engine's
    captured hipGraph must evaluate with the NEW weights afterwards (weights updated in place)."""
    import random
    from chessrl_amd import selfplay
    from chessrl_amd.agent import Agent
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.game import Game
    random.seed(1)
    np.random.seed(1)
    agent = Agent(True, blocks=2, filters=64)
    gam = selfplay.play_game(agent, max_iters=2)
    assert gam.get_result() is not None
    d = DatasetGame()
    d.loads(str(DatasetGame([gam])))                          # the JSON wire format of selfplay.py:95
    g0 = Game()
    before_pol, before_val = agent.predict(g0)
    eng = agent.engine_for(2)                                 # graph-captured engine from play_game
    hist = agent.train(d, logdir=str(tmp_path), epochs=2, validation_split=0, batch_size=1)
    assert len(hist) == 2 and hist[1]["loss"] < hist[0]["loss"]
    assert (tmp_path / "train_log.jsonl").read_text().count("\n") == 2
    after_pol, after_val = agent.predict(g0)
    assert np.abs(after_pol - before_pol).max() > 0
    assert agent.engine_for(2) is eng
    # the fused inference path and the trained fp32 tower (inference mode) agree within 1e-3
    w = agent.model.weights
    planes = encoder_oracle.get_game_state(OracleGame())[None]
    epol, eval_ = tower_oracle.forward(w, planes)
    assert np.abs(after_pol - epol[0].numpy()).max() <= 1e-3 and abs(after_val - float(eval_[0])) <= 2e-3
    # a move searched with the captured engine after training == a fresh agent with the saved weights
    path = str(tmp_path / "model-0.npz")
    agent.save(path)
    fresh = Agent(True, weights=path)
    np.random.seed(3)
    a = agent.best_move(g0, real_game=False, max_iters=2)
    np.random.seed(3)
    b = fresh.best_move(g0, real_game=False, max_iters=2)
    assert a == b
    for g in d.games + [gam, g0]:
        g.free()


@pytest.mark.gpu
def test_agent_train_on_a_json_dataset_with_validation_split(tmp_path):
    """agent.py:64-89 on a DatasetGame JSON kept slot-free: the last quarter of the games is held
    out, training continues from saved weights picked up by get_model_path."""
    import json
    from chessrl_amd.agent import Agent
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.selfplay import get_model_path
    games = []
    for s in range(8):
        g = _random_game(50 + s, 20 + 3 * s)
        games.append({"moves": g.get_history()["moves"], "result": (s % 3) - 1, "player_color": bool(s & 1),
                      "date": "02/10/2026 00:00:00"})
    games.append({"moves": [], "result": None, "player_color": True, "date": None})      # dropped on load
    data = tmp_path / "gameplays.json"
    data.write_text(json.dumps(games))
    mdir = str(tmp_path / "model")
    os.makedirs(mdir)
    ds = DatasetGame()
    ds.load(str(data), slot_free=True)
    assert len(ds) == 8
    path = get_model_path(mdir)
    assert path.endswith("model-0.npz")
    np.random.seed(0)
    agent = Agent(True, blocks=1, filters=64)
    hist = agent.train(ds, logdir=mdir, epochs=2, batch_size=2, validation_split=0.25)
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) and np.isfinite(h["val_loss"]) for h in hist)
    assert hist[1]["loss"] < hist[0]["loss"]
    agent.save(path)
    w0 = dict(np.load(path))
    again = Agent(True, weights=get_model_path(mdir))           # picks model-0.npz up and continues
    again.train(ds, epochs=1, batch_size=3, validation_split=0.25)
    again.save(path)
    w1 = dict(np.load(path))
    assert int(w1["meta.filters"]) == 64 and np.abs(w1["stem.kernel"] - w0["stem.kernel"]).max() > 0
    assert Agent(True, blocks=1, filters=64).train(DatasetGame()) is None


@pytest.mark.gpu
@pytest.mark.parametrize("boards,channels", [(1, 4), (37, 32), (380, 128)])
def test_im2col_kernels_match_the_torch_expression(boards, channels):
    """csrc/train_ops.hpp through the C-ABI: the patch matrix is an exact copy; its adjoint sums the
    same nine terms per element (order may differ from autograd's: 1e-6 relative)."""
    from chessrl_amd.train import im2col3x3
    g = torch.Generator().manual_seed(boards)
    x = torch.randn((boards, 8, 8, channels), generator=g)
    gc = torch.randn((boards * 64, 9 * channels), generator=g)
    xc = x.clone().requires_grad_(True)
    ref = im2col3x3(xc)                                   # CPU tensor: the torch expression
    ref.backward(gc)
    xd = x.cuda().requires_grad_(True)
    got = im2col3x3(xd)                                   # CUDA tensor: the HIP kernels
    got.backward(gc.cuda())
    assert torch.equal(got.cpu(), ref.detach())
    assert (xd.grad.cpu() - xc.grad).abs().max() <= 1e-6 * xc.grad.abs().max()
    # deterministic backward: bit-identical on a second run
    xd2 = x.cuda().requires_grad_(True)
    im2col3x3(xd2).backward(gc.cuda())
    assert torch.equal(xd2.grad, xd.grad)
