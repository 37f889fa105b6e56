#!/bin/bash
# round 6, GPU run 1: new tests, the in-graph step fit at C5 / C3 hybrid, rocprofv3 with one step per graph
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_measure.py -x -q > $O/test_measure.log 2>&1; echo "measure tests rc $?"
tail -3 $O/test_measure.log
timeout 120 python tools/graph_event_probe.py > $O/graph_event_probe.json 2> $O/graph_event_probe.err; echo "event probe rc $?"; cat $O/graph_event_probe.json
timeout 900 python bench.py --blocks 20 --filters 256 --steps 40 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
timeout 900 python bench.py --seed 1 --steps 400 --warmup 100 > $O/bench_c3_seed1_hybrid.json 2> $O/bench_c3_seed1.err; echo "c3 hybrid rc $?"
timeout 900 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "c3 default rc $?"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/st_c5g
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_c5g -- python3 $R/bench.py --blocks 20 --filters 256 --steps-per-graph 1 --steps 40 --warmup 10 --no-cpu-baseline --parity-positions 0 --strict-steps 0 --gph-seconds 0 > $R/$O/bench_c5_graph_profiled.json 2> $R/$O/bench_c5_graph_profiled.err; echo "rocprof c5 spg1 rc $?"
find /tmp/st_c5g -name "*kernel_stats.csv" -exec cp {} $R/$O/bench_c5_graph_kernel_stats.csv \;
ls -la $R/$O
cd $R
timeout 300 ./tools/ubench/conv_indexed 4096 10 > $O/conv_indexed_harness.log 2>&1; echo "indexed harness rc $?"; cat $O/conv_indexed_harness.log
