"""Scratch (GPU): error of the fused fp16 trunk against the fp32 oracle on TRAINED-LIKE weights.

Plays a few dozen quick self-play games, trains each BASELINE tower size on them for a few epochs
with the product's own trainer (so conv kernels, biases and BatchNorm moving statistics are what
training leaves behind, not the Keras defaults), then evaluates real positions of those games
with (a) the fused HIP trunk as shipped, (b) PyTorch emulations of precision strategies, all
against oracle/tower_oracle.py (fp32, CPU) on the same weights.

python tools/tower_trained_probe.py [epochs=4] [games=64] > gpurun_out/tower_trained.json
"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from chessrl_amd.dataset import DatasetGame
from chessrl_amd.engine import LockstepEngine
from chessrl_amd.model import ChessModel, _fold
from chessrl_amd.netencoder import DataGameSequence
from chessrl_amd.selfplay import SelfPlayRunner
from oracle import tower_oracle

dev = torch.device("cuda:0")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_games = int(sys.argv[2]) if len(sys.argv) > 2 else 64


def emulate(w, planes, mode):
    """fp16 = every layer output rounded to fp16 (PyTorch half convs); mixed = fp16 operands,
    fp32 accumulation and fp32 residual stream (what the fused kernel does); xsplit = mixed + the
    fp16 rounding error of the ACTIVATIONS fed through a second product; wsplit = mixed + the
    rounding error of the WEIGHTS; split3 = both."""
    blocks = int(w["meta.blocks"])
    x0 = torch.as_tensor(planes, dtype=torch.float32, device=dev).permute(0, 3, 1, 2).contiguous(
        memory_format=torch.channels_last)

    def conv(x32, name, bn):
        k, b = _fold(w, name, bn)
        k, b = k.to(dev), b.to(dev)
        kh, xh = k.half(), x32.half()
        y = F.conv2d(xh, kh, None, padding=1).float()
        if mode in ("xsplit", "split3"):
            y = y + F.conv2d((x32 - xh.float()).half(), kh, None, padding=1).float()
        if mode in ("wsplit", "split3"):
            y = y + F.conv2d(xh, (k - kh.float()).half(), None, padding=1).float()
        return y + b.view(1, -1, 1, 1)

    def r(t):
        return t.half().float() if mode == "fp16" else t

    x = r(conv(x0, "stem", None))
    for i in range(blocks):
        y = r(F.relu(conv(x, "block%d.conv1" % i, "block%d.bn1" % i)))
        y = r(conv(y, "block%d.conv2" % i, "block%d.bn2" % i))
        x = r(F.relu(x + y))

    def head(name, bn):
        k, b = _fold(w, name, bn)
        return F.relu(F.conv2d(x, k.to(dev), b.to(dev)))

    def t(name):
        return torch.from_numpy(np.asarray(w[name], np.float32)).to(dev)

    B = x.shape[0]
    p = head("policy.conv", "policy.bn").permute(0, 2, 3, 1).reshape(B, -1)
    p = torch.softmax(p @ t("policy.dense.kernel") + t("policy.dense.bias"), -1)
    v = head("value.conv", "value.bn").permute(0, 2, 3, 1).reshape(B, -1)
    v = F.relu(v @ t("value.dense1.kernel") + t("value.dense1.bias"))
    z = v @ t("value.dense2.kernel") + t("value.dense2.bias")
    return p.cpu(), torch.tanh(z)[:, 0].cpu()


def bn_summary(w):
    g, v, m = [], [], []
    for k in w:
        if k.endswith(".gamma") and k.startswith("block"):
            g.append(np.asarray(w[k]))
            v.append(np.asarray(w[k[:-6] + ".var"]))
            m.append(np.asarray(w[k[:-6] + ".mean"]))
    g, v, m = np.concatenate(g), np.concatenate(v), np.concatenate(m)
    gain = g / np.sqrt(v + 1e-3)
    return {"gain_min": float(gain.min()), "gain_max": float(gain.max()), "gain_mean": float(gain.mean()),
            "var_min": float(v.min()), "var_max": float(v.max()), "mean_absmax": float(np.abs(m).max())}


t0 = time.time()
play = ChessModel(blocks=6, filters=64, seed=1)
run = SelfPlayRunner(play, min(n_games, 64), 24, seed=3, noise=True, total_games=n_games, max_plies=1024)
recs = run.run()
run.close()
print("played %d games, %d plies in %.1f s" % (len(recs), sum(len(r) for r in recs), time.time() - t0),
      file=sys.stderr)
# evaluation positions: prefixes of the recorded games (real positions with history)
rng = np.random.default_rng(5)
prefixes = [list(r.moves[:int(rng.integers(0, max(1, len(r))))]) for r in recs[:48]]

out = {"epochs": epochs, "games": len(recs), "configs": []}
for blocks, filters in [(6, 64), (10, 128), (20, 256)]:
    model = ChessModel(compile_model=True, blocks=blocks, filters=filters, seed=2)
    gen = DataGameSequence(DatasetGame(list(recs)), batch_size=1, random_flips=.1)
    t0 = time.time()
    hist = model.train_generator(gen, epochs=epochs)
    w = model.weights
    eng = LockstepEngine(model, n_games=len(prefixes), max_sims=2, use_graph=False, bitplanes=False)
    eng.load_moves(prefixes)
    eng.ctx.encode(eng.planes_s1.data_ptr())
    pol, val = model(eng.planes_s1)
    planes = eng.planes_s1.float().cpu().numpy()[..., :127]
    epol, eval_ = tower_oracle.forward(w, planes)
    row = {"blocks": blocks, "filters": filters, "train_s": time.time() - t0,
           "loss": [h["loss"] for h in hist], "bn": bn_summary(w),
           "value_absmax": float(eval_.abs().max()), "policy_max": float(epol.max()),
           "fused": {"dpolicy": (pol.cpu() - epol).abs().max().item(), "dvalue": (val.cpu() - eval_).abs().max().item(),
                     "dvalue_mean": (val.cpu() - eval_).abs().mean().item()}}
    for mode in ("fp16", "mixed", "xsplit", "wsplit", "split3"):
        p, v = emulate(w, planes, mode)
        row[mode] = {"dpolicy": (p - epol).abs().max().item(), "dvalue": (v - eval_).abs().max().item()}
    eng.close()
    out["configs"].append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps(out))
