"""Scratch (GPU): time the 128-filter trunk at the C3 shape.  `python tools/trunk_bench.py [variant]`
loads the TUNING library (libchessrl_hip_tuning.so, built here with -DCRL_TUNING on first use) and
runs CRL_TRUNK_VARIANT=variant (0 = production; the ladder of profiles/r01/pmc_trunk_kernel.md;
timing-only variants give wrong results; so do all 32x32x16 variants since the weight image moved to
the x16 kernels' plane order -- this tool reports time only).  One variant per process: the library
reads it once."""
import os
import sys
os.environ["CRL_TUNING_LIB"] = "1"
os.environ["CRL_TRUNK_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd.model import ChessModel
B, blocks = 4096, 10
rng = np.random.default_rng(0)
x = torch.zeros((B, 8, 8, 128), dtype=torch.float16, device="cuda:0")
x[..., :127] = torch.from_numpy((rng.random((B, 8, 8, 127)) < 0.12).astype(np.float16)).cuda()
m = ChessModel(blocks=blocks, filters=128)
flops = 2.0 * (73152 * 128 + 1152 * 128 * 128 * blocks + 192 * 128) * B
res = []
for rnd in range(3):
    for _ in range(2):
        m._run_fused(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m._run_fused(x)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20)
t = min(res)
print("variant %s: %.3f ms (min of 3) = %.0f TFLOP/s   all=%s" % (
    os.environ["CRL_TRUNK_VARIANT"], t, flops / t / 1e9, ["%.3f" % z for z in res]))
