"""Scratch probe (GPU): error of tower precision strategies vs the fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import tower_oracle
from chessrl_amd.model import _fold, IN_PLANES

dev = torch.device("cuda:0")

def run(w, planes, mode, wq):
    blocks = int(w["meta.blocks"])
    x0 = torch.as_tensor(planes, dtype=torch.float32, device=dev).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    def conv(x32, name, bn):
        k, b = _fold(w, name, bn)
        k = k.to(dev); b = b.to(dev)
        if mode == "fp32":
            return F.conv2d(x32, k, b, padding=1)
        kh = k.half()
        xh = x32.half()
        y = F.conv2d(xh, kh, None, padding=1).float()
        if mode in ("split2", "split3"):
            xl = (x32 - xh.float()).half()
            y = y + F.conv2d(xl, kh, None, padding=1).float()
        if mode == "split3":
            kl = (k - kh.float()).half()
            y = y + F.conv2d(xh, kl, None, padding=1).float()
        return y + b.view(1, -1, 1, 1)
    x = conv(x0, "stem", None)
    if mode == "fp16": x = x.half().float()
    for i in range(blocks):
        y = F.relu(conv(x, "block%d.conv1" % i, "block%d.bn1" % i))
        if mode == "fp16": y = y.half().float()
        y = conv(y, "block%d.conv2" % i, "block%d.bn2" % i)
        if mode == "fp16": y = y.half().float()
        x = F.relu(x + y)
        if mode == "fp16": x = x.half().float()
    # heads fp32
    def head(name, bn):
        k, b = _fold(w, name, bn)
        return F.relu(F.conv2d(x, k.to(dev), b.to(dev)))
    B = x.shape[0]
    p = head("policy.conv", "policy.bn").permute(0, 2, 3, 1).reshape(B, -1)
    p = torch.softmax(p @ torch.from_numpy(w["policy.dense.kernel"]).to(dev) + torch.from_numpy(w["policy.dense.bias"]).to(dev), -1)
    v = head("value.conv", "value.bn").permute(0, 2, 3, 1).reshape(B, -1)
    v = F.relu(v @ torch.from_numpy(w["value.dense1.kernel"]).to(dev) + torch.from_numpy(w["value.dense1.bias"]).to(dev))
    z = v @ torch.from_numpy(w["value.dense2.kernel"]).to(dev) + torch.from_numpy(w["value.dense2.bias"]).to(dev)
    return p.cpu(), torch.tanh(z)[:, 0].cpu(), z[:, 0].cpu(), x.abs().max().item()

rng = np.random.default_rng(0)
planes = (rng.random((64, 8, 8, 127)) < 0.12).astype(np.float32)
for blocks, filters in [(6, 64), (10, 128), (10, 256), (20, 256)]:
    for rbn in (False, True):
        w = tower_oracle.init_weights(blocks, filters, seed=4, randomize_bn=rbn)
        ep, ev = tower_oracle.forward(w, planes)
        # oracle with fp16-rounded conv weights (after BN folding is not expressible; round raw kernels)
        line = "%2dx%-3d rbn=%d |" % (blocks, filters, rbn)
        for mode in ("fp16", "mixed", "split2", "split3", "fp32"):
            p, v, z, amax = run(w, planes, mode if mode != "mixed" else "mixed", False)
            line += " %s dv=%.2e dp=%.1e |" % (mode, (v - ev).abs().max().item(), (p - ep).abs().max().item())
        line += " |z|max=%.2f act=%.0f" % (z.abs().max().item(), amax)
        print(line, flush=True)
