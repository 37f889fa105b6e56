#!/bin/bash
# round 6, GPU run 4: the whole GPU suite + smoke on the tree with the two indexed geometries and the plane touch, bench lines,
# the round's profile pass and the SCALE-day rehearsal
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 900 python bench.py --blocks 20 --filters 256 --steps 40 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
timeout 900 python bench.py --seed 1 --steps 400 --warmup 100 > $O/bench_c3_seed1_hybrid.json 2> $O/bench_c3_seed1.err; echo "c3 hybrid rc $?"
timeout 900 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "c3 default rc $?"
ROUND=r06 timeout 2400 bash tools/profile_pass.sh > $O/profile_pass.log 2>&1; echo "profile pass rc $?"
cd $R; ROUND=r06 timeout 2400 bash tools/scale_rehearsal.sh 8 > $O/scale_rehearsal.log 2>&1; echo "rehearsal rc $?"; tail -12 $O/scale_rehearsal.log
ls $R/gpurun_out/prof_r06 | head -50
