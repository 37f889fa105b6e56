#!/bin/bash
# round 6, GPU run 6: three-pair plane ring at one board per workgroup: harness, the tower tests, the C5 line
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 ./tools/ubench/conv_indexed 4096 6 > $O/conv_indexed_harness_ring3.log 2>&1; echo "indexed harness rc $?"; cat $O/conv_indexed_harness_ring3.log
timeout 1200 python -m pytest tests/test_gpu_tower.py tests/test_gpu_measure.py -x -q > $O/test_tower_ring3.log 2>&1; echo "tower tests rc $?"; tail -3 $O/test_tower_ring3.log
timeout 900 python bench.py --blocks 20 --filters 256 --steps 40 --warmup 10 > $O/bench_c5_ring3.json 2> $O/bench_c5_ring3.err; echo "c5 rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_c5_ring3.json').read().strip().splitlines()[-1])
sf=d['roofline']['step_fit']
print(round(d['value']), round(d['ms_per_step'],3), {k:round(v['launch_ms'],4) for k,v in sf['trunk_in_step'].items()}, round(sf['ratio'],4))
PY
