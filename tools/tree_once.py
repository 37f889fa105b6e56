"""Scratch (GPU): run the search kernels for N lockstep steps with a constant evaluator (no tower
kernels), for rocprofv3 --pmc passes over k_select_expand / k_reply.
python tools/tree_once.py [games=4096] [steps=120] [bitplanes=1] [shape.json|-] [legal_priors=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.engine import LockstepEngine
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
g = torch.Generator(device="cuda:0").manual_seed(0)
pol = torch.softmax(torch.randn((G, 1968), device="cuda:0", generator=g), -1)
val = torch.tanh(torch.randn((G,), device="cuda:0", generator=g) * 0.3)


class Const(object):
    accepts_bitplanes = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0    # production: 1-KiB bitboards

    def __call__(self, planes):
        return pol, val

    accepts_legal_labels = (int(sys.argv[5]) if len(sys.argv) > 5 else 1) != 0  # production: priors of legal moves only

    def forward_into(self, planes, pol_out, val_out):      # the buffers already hold pol / val
        pass

    def forward_legal_into(self, planes, labels, counts, priors_out, val_out):
        pass


eng = LockstepEngine(Const(), G, 800, use_graph=False)
eng.pol_s1.copy_(pol); eng.pol_s2.copy_(pol); eng.val_s2.copy_(val)
if eng.legal_priors:
    eng.pri_s1.copy_(pol[:, :256]); eng.pri_s2.copy_(pol[:, 256:512])
eng.reset()
eng.search_begin()
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
c = eng.ctx.counters()
if len(sys.argv) > 4 and sys.argv[4] != "-":          # the tree shape the passes saw, for the PMC table
    import json
    c0 = dict(c)
    json.dump({"games": G, "steps": steps, "mean_depth_all": c["depth_sum"] / max(1, c["sims"]),
               "mean_branch": c["branch_sum"] / max(1, c["nodes"])}, open(sys.argv[4], "w"))
print({k: int(v) for k, v in c.items()}, "depth", c["depth_sum"] / max(1, c["sims"]), "branch", c["branch_sum"] / max(1, c["nodes"]))
