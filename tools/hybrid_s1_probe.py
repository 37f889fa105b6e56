"""Scratch (GPU): the one bounded experiment on the compliant mode, by emulation.

An evaluation of S1 (the position after our move, mctree.py:244-250) feeds ONLY an argmax over the legal
labels -- the opponent's greedy reply (agentdistributed.py:57-58).  Could S1 run the single-MFMA trunk
("f16") while S2 (priors + value: 1e-3 bar) runs "f16x3", with a fall-back to f16x3 for the boards whose
argmax is not safe?  Safe = the top-2 margin of the f16 LEGAL priors exceeds k x delta, delta = the
f16-vs-f16x3 policy distance on the probe positions of these weights (ChessModel.probe_error).

Emulated on the real population: a lockstep search of G games driven by the f16x3 tower (the truth); at
every simulation step the S1 planes + legal label lists the search kernels just wrote are ALSO evaluated
in f16, and per board we record: replies equal?  margin of the f16 priors.  Reported for k in K: the
fall-back fraction and the number of WRONG replies among the boards that would not fall back, over >= 1e6
S1 positions, for a sharp (calibrated) net and a Keras-default one.

python tools/hybrid_s1_probe.py [G=4096] [steps=300] [configs=10x128] [sims=800]
"""
import ctypes
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chessrl_amd.engine import LockstepEngine
from chessrl_amd.model import ChessModel
from oracle import tower_oracle
from tests.util import encode_prefixes, selfplay_position_prefixes

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfgs = [tuple(int(v) for v in c.split("x")) for c in (sys.argv[3] if len(sys.argv) > 3 else "10x128").split(",")]
sims = int(sys.argv[4]) if len(sys.argv) > 4 else 800
K = [1, 2, 4, 8, 16, 32, 64]


class DevArray(object):
    """a raw device pointer as something torch.as_tensor understands"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


prefixes, info = selfplay_position_prefixes(1024)
_, planes = encode_prefixes(ChessModel(blocks=2, filters=64, precision="f16"), prefixes)
out = {"games": G, "steps": steps, "sims_per_move": sims, "configs": []}
for blocks, filters in cfgs:
    for kind in ("sharp", "keras_default_init"):
        w = (tower_oracle.init_weights(blocks, filters, seed=4) if kind == "keras_default_init"
             else tower_oracle.calibrated_weights(blocks, filters, planes[:512], seed=7))
        model = ChessModel(weights=w, precision="auto")
        probe = model.probe_error()
        delta = probe["dpolicy_max"]
        model.set_precision("auto")
        # the same distance in LOG space (a rounding error of the logits moves a probability by a FACTOR):
        # max |log p16 - log p48| over the probe positions and every label that is not vanishing
        from chessrl_amd.model import _probe_bitplanes
        pp = _probe_bitplanes(model.device, model.PROBE_POSITIONS)
        pa, _ = model._forward_fused(pp, precision="f16")
        pb, _ = model._forward_fused(pp, precision="f16x3")
        ok = pb > 1e-12
        delta_log = float((pa[ok].log() - pb[ok].log()).abs().max())
        truth = "f16x3"
        if model.precision != truth:
            model.precision = truth                      # the search is driven by the fp32-grade mode
        eng = LockstepEngine(model, G, sims, use_graph=False, raw_priors=False)
        eng.reset()
        # spread the games over the game: random playouts of 0..119 plies first
        rng = np.random.RandomState(11)
        target = rng.randint(0, 120, size=G)
        for ply in range(int(target.max())):
            moves, counts = eng.ctx.legal_moves()
            pick = (rng.random_sample(G) * np.maximum(counts, 1)).astype(np.int64)
            mv = np.where((target > ply) & (counts > 0), moves[np.arange(G), pick], 0xFFFF).astype(np.uint16)
            eng.ctx.push_moves(mv)
        eng.search_begin()
        lab_ptr, cnt_ptr = eng._lab_s1
        cnt = torch.as_tensor(DevArray(cnt_ptr, (G,), "<i4"), device="cuda:0")
        pri16 = torch.zeros((G, 256), dtype=torch.float32, device="cuda:0")
        cols = torch.arange(256, device="cuda:0")[None, :]
        n_pos = n_diff = 0
        margins, wrong_margins, lmargins, wrong_lmargins = [], [], [], []
        t0 = time.time()
        for s in range(steps):
            eng.phase_select_expand()
            # the S1 boards of this step in BOTH modes (legal priors, normalised: the one-pass / sliced heads)
            model.precision = "f16"
            model.forward_legal_into(eng.planes_s1, lab_ptr, cnt_ptr, pri16, None)
            model.precision = truth
            eng.phase_tower_s1()
            live = cnt > 0
            mask = cols < cnt[:, None]
            a = torch.where(mask, pri16, torch.full_like(pri16, -1.0))
            b = torch.where(mask, eng.pri_s1, torch.full_like(pri16, -1.0))
            top2 = a.topk(2, dim=1).values
            margin = top2[:, 0] - torch.clamp(top2[:, 1], min=0.0)
            lmargin = torch.where(top2[:, 1] > 0, (top2[:, 0] / top2[:, 1].clamp(min=1e-38)).log(),
                                  torch.full_like(margin, 1e9))     # a single legal move: nothing to confuse
            differ = (a.argmax(1) != b.argmax(1)) & live
            n_pos += int(live.sum())
            n_diff += int(differ.sum())
            margins.append(margin[live].cpu().numpy())
            lmargins.append(lmargin[live].cpu().numpy())
            if bool(differ.any()):
                wrong_margins.append(margin[differ].cpu().numpy())
                wrong_lmargins.append(lmargin[differ].cpu().numpy())
            eng.phase_reply()
            eng.phase_tower_s2()
        eng.ctx.sync()
        eng.close()
        margins = np.concatenate(margins)
        wrong = np.concatenate(wrong_margins) if wrong_margins else np.zeros(0)
        lmargins = np.concatenate(lmargins)
        lwrong = np.concatenate(wrong_lmargins) if wrong_lmargins else np.zeros(0)
        rec = {"blocks": blocks, "filters": filters, "weights": kind, "probe": probe, "s1_positions": int(n_pos),
               "replies_that_differ_f16_vs_f16x3": int(n_diff), "seconds": time.time() - t0,
               "largest_margin_of_a_differing_reply": float(wrong.max()) if len(wrong) else 0.0,
               "rule": [{"k": k, "threshold": k * delta, "fallback_fraction": float((margins < k * delta).mean()),
                         "wrong_replies_not_caught": int((wrong >= k * delta).sum())} for k in K],
               "probe_log_distance": delta_log,
               "largest_log_margin_of_a_differing_reply": float(lwrong.max()) if len(lwrong) else 0.0,
               "log_rule": [{"k": k, "threshold": k * delta_log,
                             "fallback_fraction": float((lmargins < k * delta_log).mean()),
                             "wrong_replies_not_caught": int((lwrong >= k * delta_log).sum())} for k in (0.5, 1, 2, 4, 8)]}
        print(json.dumps(rec), flush=True)
        out["configs"].append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/hybrid_s1_probe.json", "w"), indent=1)
