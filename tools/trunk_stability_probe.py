"""Scratch (GPU): are the fused trunk kernels' outputs identical from launch to launch while ANOTHER
process keeps the GPU busy (its workgroups share CUs and LDS bandwidth with ours, so LDS latencies
jitter)?  A kernel that touches a register before its hand-counted wait has made it valid passes every
single-process test and fails here (round 3: the 64-filter split-precision kernels, 45-90 % of the
launches wrong; cause and static check: tools/check_asm_hazards.py).
    python tools/trunk_stability_probe.py disturb 400 &  python tools/trunk_stability_probe.py measure 150 [BxFxN,...]
A second disturber, ``stream N``, delays the OTHER side of the pipeline: it streams two 4-GiB buffers through
the L2s and HBM (copies, N times), so that the weight tiles' LDS-DMA (L2 -> LDS) arrives late and a ring slot
read before its counted ``vmcnt`` wait + barrier shows (static counterpart: tools/lds_race_check.py)."""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chessrl_amd import _lib, model as M
from chessrl_amd.model import ChessModel
role = sys.argv[1]
md = lambda t: hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:6]
if role == "disturb":                     # what the other rank's start-up does: contexts and probes come and go
    for rep in range(int(sys.argv[2])):
        M._PROBE.clear()
        M._probe_bitplanes(torch.device("cuda:0"), 256)
    print("disturber done", flush=True)
elif role == "stream":                    # a bandwidth hog: 8 GiB of traffic per iteration through every L2 channel
    a = torch.empty(1 << 30, dtype=torch.float32, device="cuda:0")          # 4 GiB
    b = torch.empty(1 << 30, dtype=torch.float32, device="cuda:0")
    for rep in range(int(sys.argv[2])):
        b.copy_(a)
        a.copy_(b)
        if rep % 16 == 15:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("streamer done", flush=True)
else:
    reps = int(sys.argv[2])
    shapes = ((1, 64, 256), (1, 64, 2048), (2, 128, 256), (2, 128, 2048), (1, 256, 256), (1, 256, 1024))
    if len(sys.argv) > 3:                 # e.g. 4x256x4096: the layer-wise kernels over full rounds of workgroups
        shapes = tuple(tuple(int(x) for x in sh.split("x")) for sh in sys.argv[3].split(","))
    for blocks, filters, n in shapes:
        m = ChessModel(blocks=blocks, filters=filters, seed=5, precision="f16")
        planes = M._probe_bitplanes(m.device, 256).repeat(n // 256, 1).contiguous()
        m.precision_requested = "auto"; m._pack_fused(m.weights)
        for mode in ("f16", "f16x3"):
            seen = {}
            for rep in range(reps):
                _, hp = m._run_fused(planes, precision=mode)
                torch.cuda.synchronize()
                h = md(hp)
                seen[h] = seen.get(h, 0) + 1
            print("%dx%d %4d boards %-6s" % (blocks, filters, n, mode), "STABLE" if len(seen) == 1 else "UNSTABLE", seen, flush=True)
