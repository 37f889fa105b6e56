#!/bin/bash
# The round's profile passes on the GPU box (run through gpurun): ROUND=r03 bash tools/profile_pass.sh.
# Writes raw rocprofv3 output under gpurun_out/prof_$ROUND/; tools/distill_profiles.py turns it into
# profiles/$ROUND/ and profiles/pmc_traffic.json.  PMC passes carry --kernel-trace only (no other trace
# domain); the program after -- is python3 itself.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r06}
O=$R/gpurun_out/prof_$ROUND
mkdir -p $O
pmc() {   # name counter cmd...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pm_$name
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm_$name -- "$@" > /dev/null 2>&1
  find /tmp/pm_$name -name "*counter_collection.csv" -exec cp {} $O/${name}_${ctr}.csv \;
}
pmc trunk128 FETCH_SIZE python3 $R/tools/trunk_once.py 10 128 4096
pmc trunk128 WRITE_SIZE python3 $R/tools/trunk_once.py 10 128 4096
pmc trunk256 FETCH_SIZE python3 $R/tools/trunk_once.py 20 256 4096
pmc trunk256 WRITE_SIZE python3 $R/tools/trunk_once.py 20 256 4096
pmc trunk64 FETCH_SIZE python3 $R/tools/trunk_once.py 6 64 512
pmc trunk64 WRITE_SIZE python3 $R/tools/trunk_once.py 6 64 512
pmc trunk128x3 FETCH_SIZE python3 $R/tools/trunk_once.py 10 128 4096 f16x3
pmc trunk128x3 WRITE_SIZE python3 $R/tools/trunk_once.py 10 128 4096 f16x3
pmc trunk256x3 FETCH_SIZE python3 $R/tools/trunk_once.py 20 256 4096 f16x3
pmc trunk256x3 WRITE_SIZE python3 $R/tools/trunk_once.py 20 256 4096 f16x3
# SQ pass (one run, four SQ counters -- the block has 8 slots -- + GRBM_GUI_ACTIVE for the clock) over the three C3 kernels auto / hybrid dispatch and the layer-wise
# kernels of C5's compliant mode: MFMA busy cycles, wave cycles, LDS instructions, LDS bank conflicts
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
sq() {   # name cmd...
  local name=$1; shift
  rm -rf /tmp/sq_$name
  timeout 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/sq_$name -- "$@" > /dev/null 2>&1
  find /tmp/sq_$name -name "*counter_collection.csv" -exec cp {} $O/${name}_SQ.csv \;
}
sq trunk128 python3 $R/tools/trunk_once.py 10 128 4096
sq trunk128x3 python3 $R/tools/trunk_once.py 10 128 4096 f16x3
sq trunk128idx python3 $R/tools/trunk_once.py 10 128 4096 indexed
sq trunk256x3 python3 $R/tools/trunk_once.py 20 256 4096 f16x3
sq trunk256 python3 $R/tools/trunk_once.py 20 256 4096
pmc tree FETCH_SIZE python3 $R/tools/tree_once.py 4096 400 1 $O/tree_shape.json 1
pmc tree WRITE_SIZE python3 $R/tools/tree_once.py 4096 400 1 - 1
pmc treefull FETCH_SIZE python3 $R/tools/tree_once.py 4096 400 1 $O/treefull_shape.json 0
pmc treefull WRITE_SIZE python3 $R/tools/tree_once.py 4096 400 1 - 0
# (c5: --no-graph.  rocprofv3's kernel tracing of ROCm 7.2 does not survive a hipGraph launch that holds the layer-wise kernels:
# "AQL packet is malformed" / SIGSEGV in a tracer thread, at ONE step per graph (~135 nodes) as at eight.  It is not the node count
# (graphs of 1024 plain kernel nodes trace fine, so do C3 hybrid steps and C5 f16 steps) and not the private segment those kernels
# had in round 5 (round 6's are scratch-free and die the same way): profiles/r06.  "c5g" is that attempt, kept so that a later
# stack shows at once when it starts working; eager launches run the same kernels, and the in-graph times of a C5 step come from
# bench.py's stamp kernels (roofline.step_fit))
for cfg in "c3 --steps 800 --warmup 200" "c5g --blocks 20 --filters 256 --steps 40 --warmup 10 --steps-per-graph 1 --strict-steps 0" "c5 --blocks 20 --filters 256 --steps 100 --warmup 20 --no-graph" "c2 --games 512 --sims 100 --blocks 6 --filters 64 --steps 1600 --warmup 200"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/st_$name
  lim=900; [ "$name" = c5g ] && lim=400        # (a tracer that aborts the queue leaves the process hanging)
  timeout $lim rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- python3 $R/bench.py --no-cpu-baseline --parity-positions 0 --gph-seconds 0 "$@" > $O/bench_${name}_profiled.json 2> $O/bench_${name}.err
  find /tmp/st_$name -name "*kernel_stats.csv" -exec cp {} $O/bench_${name}_kernel_stats.csv \;
  find /tmp/st_$name -name "*domain_stats.csv" -exec cp {} $O/bench_${name}_domain_stats.csv \;
done
ls -la $O
