"""The RCCL branch of the multi-GPU path, executed on ONE GPU.

RCCL refuses two ranks on one device, and the pool's boxes have one GPU: a communicator of ONE
rank on cuda:0 still runs everything the 8-GPU job runs on each rank -- ``init_process_group("nccl",
device_id=...)`` exactly as bench.py / the CLI call it, the two ``all_gather_into_tensor`` calls and
device-to-host copies of ``records.gather_blocks`` (its world == 1 shortcut bypassed), the flat
weight broadcast of ``train.broadcast_weights`` and the bench's result reductions.  Runs in a child
process: a process group is process-global state."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
from chessrl_amd import records
from chessrl_amd.model import init_weights
from chessrl_amd.train import broadcast_weights
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
t = torch.tensor([3.0, 5.0], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
rng = np.random.default_rng(0)
n, max_plies = 37, 512
plies = rng.integers(0, max_plies + 1, n)
moves = rng.integers(0, 4096, (n, max_plies)).astype(np.uint16)
block = records.pack_arrays(5 + 8 * np.arange(n), moves, plies, rng.integers(-1, 3, n),
                            rng.integers(0, 2, n), max_plies)
st = {}
rows, counts = records.gather_blocks(block, stats=st, force_collective=True)
w = records.HEADER + (int(plies.max()) + 1) // 2            # the blocks travel trimmed to the longest record
assert counts == [n] and np.array_equal(rows, block[:, :w]), "all_gather of one rank must return the block"
assert st["backend"] == "nccl" and st["bytes_gathered"] == n * w * 4
recs = records.unpack(rows)
assert [len(r.moves) for r in recs] == list(plies)
w = init_weights(2, 32, seed=3)
w2 = broadcast_weights(w, "cuda:0", src=0)
assert all(np.array_equal(w[k], w2[k]) for k in w)
dist.destroy_process_group()
print(json.dumps({"ok": True, "gather_ms": st["ms"], "rows": int(rows.shape[0]), "t": t.tolist()}))
"""


def test_one_rank_rccl_communicator_runs_the_record_gather_and_the_weight_broadcast():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29653", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["rows"] == 37 and res["t"] == [3.0, 5.0]


def test_bench_one_rank_under_a_launcher_environment_takes_the_rccl_branch():
    """bench.py as the driver's launcher starts it (RANK / WORLD_SIZE / MASTER_* in the environment)
    with a world of ONE: ``CRL_BENCH_FORCE_GROUP=1`` makes it create the RCCL process group and time
    the record gather anyway, so init_ranks' nccl branch, the reductions and ``record_gather`` run
    on hardware."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29654", CRL_BENCH_FORCE_GROUP="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6",
                          "--warmup", "2", "--games", "64", "--sims", "16", "--blocks", "2", "--filters", "64",
                          "--no-cpu-baseline", "--gph-seconds", "3"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["value"] > 0
    # round 6's fields: the games left their common line before the window; games/hour is MEASURED in the run
    assert res["window"]["distinct_root_positions"] > 32 and res["window"]["games"] == 64
    gph = res["self_play_games_per_hour_measured"]
    assert gph["seconds"] > 0 and gph["games"] >= 0 and gph["simulations_per_s"] > 0 and "C2" in gph["config"]
    g = res["record_gather"]
    assert g["backend"] == "nccl" and g["records"] == 64 and g["ms"] > 0
