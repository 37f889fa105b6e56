#!/bin/bash
# round 6, GPU run 2: fixed tests, diverse-opening bench lines (C5 / C3 hybrid / C3 default / C2), tracer node-count probe,
# staggered persistent tower harness
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_measure.py -x -q > $O/test_measure.log 2>&1; echo "measure tests rc $?"; tail -3 $O/test_measure.log
timeout 900 python -m pytest tests/test_gpu_tower.py -x -q -k "reply_margin_lists or hybrid_mode_searches or reload_under" > $O/test_tower_idx.log 2>&1; echo "tower idx tests rc $?"; tail -3 $O/test_tower_idx.log
timeout 900 python bench.py --blocks 20 --filters 256 --steps 40 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
timeout 900 python bench.py --seed 1 --steps 400 --warmup 100 > $O/bench_c3_seed1_hybrid.json 2> $O/bench_c3_seed1.err; echo "c3 hybrid rc $?"
timeout 900 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "c3 default rc $?"
timeout 900 python bench.py --games 512 --sims 100 --blocks 6 --filters 64 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc $?"
timeout 600 ./tools/ubench/tower_persist 4096 20 5 > $O/tower_persist_stagger.log 2>&1; echo "persist harness rc $?"; cat $O/tower_persist_stagger.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gt -- python3 $R/tools/graph_trace_probe.py > /dev/null 2> $R/$O/graph_trace_probe.log; echo "trace probe rc $?"; grep "graph of\|survived" $R/$O/graph_trace_probe.log | tail -4
ls -la $R/$O
