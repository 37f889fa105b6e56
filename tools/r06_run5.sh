#!/bin/bash
# round 6, GPU run 5: the tree with scratch-free layer-wise kernels (asm MFMAs): GPU suite + smoke, bench lines, profile pass
# (incl. the graph-traced C5 pass), harnesses
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 300 ./tools/ubench/conv_layer 4096 20 > $O/conv_layer_harness_asm_mfma.log 2>&1; echo "conv_layer harness rc $?"; grep -v "^proxy" $O/conv_layer_harness_asm_mfma.log
timeout 900 python bench.py --blocks 20 --filters 256 --steps 40 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
timeout 900 python bench.py --seed 1 --steps 400 --warmup 100 > $O/bench_c3_seed1_hybrid.json 2> $O/bench_c3_seed1.err; echo "c3 hybrid rc $?"
timeout 900 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "c3 default rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_shape.json 2> $O/bench_c3_driver_shape.err; echo "c3 driver shape rc $?"
timeout 900 python bench.py --games 512 --sims 100 --blocks 6 --filters 64 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc $?"
ROUND=r06 timeout 2700 bash tools/profile_pass.sh > $O/profile_pass.log 2>&1; echo "profile pass rc $?"
ls -la $R/gpurun_out/prof_r06 | grep "c5g\|c5_"
