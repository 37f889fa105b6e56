#!/bin/bash
# round 6, GPU run 13: the hybrid bench line with the reply-rule check on it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python bench.py --seed 1 --steps 400 --warmup 100 > $O/bench_c3_seed1_hybrid.json 2> $O/bench_c3_seed1.err; echo "c3 hybrid rc $?"
timeout 900 python bench.py --blocks 20 --filters 256 --steps 40 --warmup 10 --gph-seconds 0 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
python - <<'PY'
import json
for f in ('bench_c3_seed1_hybrid.json','bench_c5.json'):
    d=json.loads(open('gpurun_out/r06/'+f).read().strip().splitlines()[-1])
    print(f, round(d['value']), d['precision_modes']['hybrid'].get('reply_rule_on_the_last_leaves'), round(d['roofline']['step_fit']['ratio'],4))
PY
