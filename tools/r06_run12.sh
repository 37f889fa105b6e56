#!/bin/bash
# round 6, GPU run 12: host-side change (run-time check of hybrid's reply rule): GPU suite again, a whole C3 round in hybrid
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -3 $O/pytest_gpu.log
CRL_SEED=1 CRL_TAG=_hybrid_seed1 timeout 1800 python tools/rolling_probe.py 4096 800 1 4096 notrain > $O/rolling_probe_hybrid.log 2>&1; echo "rolling hybrid rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/rolling_probe_hybrid_seed1.json'))
print(round(d['games_per_hour_overall']), round(d['seconds_total'],1), d['tower_precision_guard'])
PY
