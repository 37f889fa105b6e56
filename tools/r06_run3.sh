#!/bin/bash
# round 6, GPU run 3: indexed harness with cold planes, sub-populations on two streams in the hybrid mode, which feature of a
# C5 hybrid graph kills rocprofv3's kernel tracing (graphs of 1024 plain kernel nodes survive: run 2)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 ./tools/ubench/conv_indexed 4096 6 > $O/conv_indexed_harness_cold.log 2>&1; echo "indexed harness rc $?"; cat $O/conv_indexed_harness_cold.log
timeout 1200 python tools/overlap_hybrid_probe.py 4096 20 256 800 1,2 48 hybrid > $O/overlap_c5.log 2>&1; echo "overlap c5 rc $?"; grep "^{" $O/overlap_c5.log
timeout 600 python tools/overlap_hybrid_probe.py 4096 10 128 800 1,2 96 hybrid > $O/overlap_c3.log 2>&1; echo "overlap c3 rc $?"; grep "^{" $O/overlap_c3.log
cd /tmp && export TMPDIR=/tmp
Q="--steps 16 --warmup 8 --no-cpu-baseline --parity-positions 0 --strict-steps 0 --gph-seconds 0 --opening-moves 2"
tr() {  # name args...
  local name=$1; shift
  rm -rf /tmp/tr_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$name -- python3 $R/bench.py $Q "$@" > $R/$O/trace_$name.json 2> $R/$O/trace_$name.err
  echo "trace $name rc $?"
  find /tmp/tr_$name -name "*kernel_stats.csv" -exec cp {} $R/$O/trace_${name}_kernel_stats.csv \;
}
tr c3_hybrid_spg8 --seed 1
tr c3_hybrid_spg1 --seed 1 --steps-per-graph 1
tr c5_f16x3_spg1 --blocks 20 --filters 256 --precision f16x3 --steps-per-graph 1
tr c5_f16_spg8 --blocks 20 --filters 256 --precision f16
tr c5_hybrid_spg1_nostamps --blocks 20 --filters 256 --steps-per-graph 1 --graph-phase-steps 0
ls -la $R/$O | tail -30
