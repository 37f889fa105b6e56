#!/bin/bash
# SCALE-day rehearsal on ONE GPU: eight rank processes of bench.py with real GPU work share device 0 over a
# gloo process group -- the whole multi-rank control flow of the bench: precision-mode agreement, the parity gate on rank 0
# while seven ranks wait in a collective (explicit process-group timeout), the re-timing branch (forced: a SHARP weight file
# timed in --precision f16 misses the 1e-3 bar and every rank times again in the compliant mode), record_gather, teardown.
# Once through bench.py's own launcher (children of a parent that never touches the GPU) and once under
# torch.distributed.run (the driver's form; the launcher starts before any GPU call).   bash tools/scale_rehearsal.sh [ranks=8]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-8}
O=$R/gpurun_out/rehearsal_${ROUND:-r06}
mkdir -p $O
cd $R
export CRL_BENCH_DEVICE=0 CRL_BENCH_BACKEND=gloo
C2="--games 512 --sims 100 --blocks 6 --filters 64 --steps 200 --warmup 50 --cpu-seconds 5"
python tools/make_sharp_weights.py $O/sharp_6x64.npz 6 64 7 > $O/make_sharp.log 2>&1
echo "== self-spawn, random init (auto)"; timeout 900 python bench.py --gpus $N $C2 > $O/bench_${N}rank_gloo_one_gpu.json 2> $O/bench_${N}rank_gloo_one_gpu.err; echo rc=$?
echo "== self-spawn, sharp weights timed in f16 -> re-timed"; timeout 900 python bench.py --gpus $N $C2 --precision f16 --weights $O/sharp_6x64.npz > $O/bench_${N}rank_gloo_one_gpu_retimed.json 2> $O/bench_${N}rank_gloo_one_gpu_retimed.err; echo rc=$?
echo "== torch.distributed.run, sharp weights timed in f16 -> re-timed"; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N $C2 --precision f16 --weights $O/sharp_6x64.npz > $O/bench_${N}rank_torchrun_one_gpu_retimed.json 2> $O/bench_${N}rank_torchrun_one_gpu_retimed.err; echo rc=$?
echo "== self-spawn, C4's size: eight full C3 shards (8 x 4096 games, 800 sims/move, 10x128) on one GPU"; timeout 1500 python bench.py --gpus $N --steps 40 --warmup 10 --cpu-seconds 5 --gph-seconds 0 > $O/bench_${N}rank_gloo_one_gpu_c4_shards.json 2> $O/bench_${N}rank_gloo_one_gpu_c4_shards.err; echo rc=$?
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step")}, d["config"]["tower_precision"], d["config"]["tower_precision_why"][:60],
          "gather:", d.get("record_gather", {}).get("records"), d.get("record_gather", {}).get("backend"),
          "parity:", (d.get("tower_error_vs_fp32") or {}).get("within_bar"), (d.get("tower_error_vs_fp32") or {}).get("first_timed_mode"))
except Exception as e:
    print("no line:", e)
PY
done
