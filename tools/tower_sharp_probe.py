"""Scratch (GPU): the 1e-3 tower bar on positions harvested from real self-play, for two kinds of
weights -- Keras-default init (what bench.py times) and a SENSITIVE net (unit gain per layer, peaked
policy, spread value: oracle/tower_oracle.calibrated_weights) -- and the precision modes of the
fused trunk ("f16": one MFMA per product; "f16x3": split operands, three MFMAs) plus the PyTorch-ROCm
f32 path, against the fp32 CPU oracle; what ``precision="auto"`` decides; kernel times per mode.
python tools/tower_sharp_probe.py [n_positions=4096] [configs e.g. 6x64,10x128]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chessrl_amd.model import ChessModel
from oracle import tower_oracle
from tests.util import encode_prefixes, selfplay_position_prefixes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfgs = [tuple(int(v) for v in c.split("x")) for c in (sys.argv[2] if len(sys.argv) > 2 else "6x64,10x128,10x256,20x256").split(",")]
prefixes, info = selfplay_position_prefixes(n)
out = {"positions": n, "harvest": info, "configs": []}
x_bits, planes = encode_prefixes(ChessModel(blocks=2, filters=64, precision="f16"), prefixes)


def ktime(model, mode):
    for _ in range(3):
        model._run_fused(x_bits, precision=mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        model._run_fused(x_bits, precision=mode)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


for blocks, filters in cfgs:
    for kind in ("keras_default_init", "sharp"):
        t0 = time.time()
        w = (tower_oracle.init_weights(blocks, filters, seed=4) if kind == "keras_default_init"
             else tower_oracle.calibrated_weights(blocks, filters, planes[:512], seed=7))
        epol, eval_ = tower_oracle.forward(w, planes)
        rec = {"blocks": blocks, "filters": filters, "weights": kind, "oracle_s": time.time() - t0,
               "policy_max": float(epol.max()), "policy_max_mean": float(epol.max(1).values.mean()),
               "value_absmax": float(eval_.abs().max()), "value_std": float(eval_.std())}
        auto = ChessModel(weights=w, precision="auto")
        rec["auto"] = auto.precision_probe
        for name in ("f16", "f16x3", "torch_f32"):
            if name == "torch_f32":
                model = ChessModel(weights=w, dtype=torch.float32, fused=False)
                xin = torch.zeros((n, 8, 8, 128), dtype=torch.float32, device="cuda:0")
                xin[..., :127] = torch.from_numpy(planes).cuda()
                pol, val = model(xin)
                ms = None
            else:
                pol, val = auto._forward_fused(x_bits, precision=name)
                ms = ktime(auto, name)
            dp = (pol.cpu() - epol).abs().max(1).values.numpy()
            dv = (val.cpu() - eval_).abs().numpy()
            rec[name] = {"dpolicy_max": float(dp.max()), "dpolicy_p999": float(np.quantile(dp, 0.999)),
                         "dpolicy_median": float(np.median(dp)),
                         "dvalue_max": float(dv.max()), "dvalue_p999": float(np.quantile(dv, 0.999)),
                         "dvalue_median": float(np.median(dv)),
                         "over_1e-3": int(((dp > 1e-3) | (dv > 1e-3)).sum()), "trunk_kernel_ms": ms}
        print(json.dumps(rec), flush=True)
        out["configs"].append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/tower_sharp_probe.json", "w"), indent=1)
print(json.dumps(out["harvest"]))
