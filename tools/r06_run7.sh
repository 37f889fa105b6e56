#!/bin/bash
# round 6, GPU run 7 (the final tree): GPU suite + smoke, the bench lines, whole-game probes (rolling rounds), SCALE-day rehearsal
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "c3 default rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_shape.json 2> $O/bench_c3_driver_shape.err; echo "c3 driver shape rc $?"
timeout 900 python bench.py --seed 1 --steps 400 --warmup 100 > $O/bench_c3_seed1_hybrid.json 2> $O/bench_c3_seed1.err; echo "c3 hybrid rc $?"
timeout 900 python bench.py --blocks 20 --filters 256 --steps 40 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
timeout 900 python bench.py --games 512 --sims 100 --blocks 6 --filters 64 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc $?"
timeout 1500 python tools/rolling_probe.py 4096 800 2 4096 notrain > $O/rolling_probe_f16.log 2>&1; echo "rolling f16 rc $?"; tail -1 $O/rolling_probe_f16.log | cut -c1-300
CRL_SEED=1 CRL_TAG=_hybrid_seed1 timeout 1800 python tools/rolling_probe.py 4096 800 1 4096 notrain > $O/rolling_probe_hybrid.log 2>&1; echo "rolling hybrid rc $?"; tail -1 $O/rolling_probe_hybrid.log | cut -c1-300
ROUND=r06 timeout 2400 bash tools/scale_rehearsal.sh 8 > $O/scale_rehearsal.log 2>&1; echo "rehearsal rc $?"; tail -9 $O/scale_rehearsal.log | cut -c1-200
