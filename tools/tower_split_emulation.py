"""Scratch (CPU): WHERE the fp16 error of the fused trunk comes from on a sensitive net, by emulation.
The trunk kernel rounds the OPERANDS of every convolution (activations and BN-folded weights) to fp16
and accumulates in fp32 with an fp32 skip stream; here the same rounding is applied in fp32 torch, per
layer and per operand, on the calibrated ("sharp") weights of oracle/tower_oracle.calibrated_weights:
  all       every conv rounds both operands (= the f16 kernel)
  last1/2   only the LAST block (+ head convs) keeps full precision (the lever VERDICT r2 suggested)
  first10   the first half of the convs keep full precision
  x only / w only   one operand keeps full precision everywhere
  x3        both operands split x = hi + lo, product = hi.hi + hi.lo + lo.hi (three fp16 MFMAs)
python tools/tower_split_emulation.py [blocks=10] [filters=128] [positions=512]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import tower_oracle as T
from oracle.chess_oracle import OracleGame, move_to_uci
from oracle import encoder_oracle

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
filters = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
rng = np.random.default_rng(0)
planes = []
while len(planes) < n:                       # random playouts from the start position (CPU oracle rules)
    g = OracleGame()
    for ply in range(int(rng.integers(0, 160))):
        lm = g.legal_move_ids()
        if len(lm) == 0 or g.get_result() is not None:
            break
        g.move(move_to_uci(lm[int(rng.integers(len(lm)))]))
    planes.append(encoder_oracle.get_game_state(g))
planes = np.stack(planes).astype(np.float32)
w = T.calibrated_weights(blocks, filters, planes[:256], seed=7)
epol, eval_ = T.forward(w, planes)


def h(x):
    return x.half().float()


def fold(w, conv, bn):
    k = torch.from_numpy(w[conv + ".kernel"]).permute(3, 2, 0, 1).contiguous()
    b = torch.from_numpy(w[conv + ".bias"]).clone()
    if bn:
        s = torch.from_numpy(w[bn + ".gamma"]) / torch.sqrt(torch.from_numpy(w[bn + ".var"]) + T.BN_EPS)
        k = k * s.view(-1, 1, 1, 1)
        b = (b - torch.from_numpy(w[bn + ".mean"])) * s + torch.from_numpy(w[bn + ".beta"])
    return k, b


@torch.no_grad()
def emulate(round_x, round_w, x3=()):
    """round_x / round_w: sets of conv indices (0 = stem, 1.. = block convs) whose activation / weight
    operand is rounded to fp16; x3: conv indices computed with the three-product split instead."""
    names = [("stem", None)]
    for i in range(blocks):
        names += [("block%d.conv1" % i, "block%d.bn1" % i), ("block%d.conv2" % i, "block%d.bn2" % i)]
    x = torch.from_numpy(planes).permute(0, 3, 1, 2)
    res = None
    for ci, (conv, bn) in enumerate(names):
        k, b = fold(w, conv, bn)
        if ci in x3:
            xh, kh = h(x), h(k)
            xl, kl = h(x - xh), h(k - kh)
            y = F.conv2d(xh, kh, b, padding=1) + F.conv2d(xh, kl, None, padding=1) + F.conv2d(xl, kh, None, padding=1)
        else:
            y = F.conv2d(h(x) if ci in round_x else x, h(k) if ci in round_w else k, b, padding=1)
        if ci == 0:
            res = y
            x = y
        elif ci % 2 == 1:
            x = F.relu(y)
        else:
            res = F.relu(res + y)
            x = res
    feat = res
    p = F.relu(T._bn(T._conv(feat, w, "policy.conv", 0), w, "policy.bn"))
    p = p.permute(0, 2, 3, 1).reshape(p.shape[0], -1)
    p = torch.softmax(p @ torch.from_numpy(w["policy.dense.kernel"]) + torch.from_numpy(w["policy.dense.bias"]), -1)
    v = F.relu(T._bn(T._conv(feat, w, "value.conv", 0), w, "value.bn"))
    v = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
    v = F.relu(v @ torch.from_numpy(w["value.dense1.kernel"]) + torch.from_numpy(w["value.dense1.bias"]))
    v = torch.tanh(v @ torch.from_numpy(w["value.dense2.kernel"]) + torch.from_numpy(w["value.dense2.bias"]))[:, 0]
    return float((p - epol).abs().max()), float((v - eval_).abs().max())


nc = 1 + 2 * blocks
allc = set(range(nc))
cases = {
    "all convs fp16 operands (= the f16 kernel)": (allc, allc, ()),
    "last block full precision": (set(range(nc - 2)), set(range(nc - 2)), ()),
    "last 2 blocks full precision": (set(range(nc - 4)), set(range(nc - 4)), ()),
    "first half of the convs full precision": (set(range(nc // 2, nc)), set(range(nc // 2, nc)), ()),
    "activations full precision, weights fp16": (set(), allc, ()),
    "weights full precision, activations fp16": (allc, set(), ()),
    "x3 split in the last block only": (set(range(nc - 2)), set(range(nc - 2)), (nc - 2, nc - 1)),
    "x3 split in every conv": (set(), set(), tuple(range(nc))),
}
out = {"blocks": blocks, "filters": filters, "positions": n, "policy_max": float(epol.max()),
       "value_absmax": float(eval_.abs().max()), "cases": {}}
for name, (rx, rw, x3) in cases.items():
    dp, dv = emulate(rx, rw, x3)
    out["cases"][name] = {"dpolicy_max": dp, "dvalue_max": dv}
    print("%-48s dpolicy %.2e  dvalue %.2e" % (name, dp, dv), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/tower_split_emulation_%dx%d.json" % (blocks, filters), "w"), indent=1)
