#!/bin/bash
# round 6, GPU run 14: the compliant mode on SHARP nets (what a trained net looks like to the precision modes) on games that differ:
# how many S1 boards the margin lists, what the step costs, whether the reply rule holds (C3 10x128 and C5 20x256)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/make_sharp_weights.py $O/sharp_10x128.npz 10 128 7 > $O/make_sharp.log 2>&1
python tools/make_sharp_weights.py $O/sharp_20x256.npz 20 256 7 >> $O/make_sharp.log 2>&1
timeout 900 python bench.py --weights $O/sharp_10x128.npz --steps 400 --warmup 100 --gph-seconds 0 --no-cpu-baseline > $O/bench_c3_sharp.json 2> $O/bench_c3_sharp.err; echo "c3 sharp rc $?"
timeout 900 python bench.py --weights $O/sharp_20x256.npz --steps 40 --warmup 10 --gph-seconds 0 --no-cpu-baseline > $O/bench_c5_sharp.json 2> $O/bench_c5_sharp.err; echo "c5 sharp rc $?"
rm -f $O/sharp_10x128.npz $O/sharp_20x256.npz
python - <<'PY'
import json
for f in ('bench_c3_sharp.json','bench_c5_sharp.json'):
    d=json.loads(open('gpurun_out/r06/'+f).read().strip().splitlines()[-1])
    h=d['precision_modes']['hybrid']; sf=d['roofline']['step_fit']
    print(f, d['config']['tower_precision'], round(d['value']), round(d['ms_per_step'],3), 'twice', h.get('s1_boards_evaluated_twice'), 'rule', h.get('reply_rule_on_the_last_leaves'), 'margin', h.get('reply_margin'), {k:round(v['launch_ms'],3) for k,v in sf['trunk_in_step'].items()}, round(sf['ratio'],4), 'parity', {k:d['tower_error_vs_fp32'][k] for k in ('dpolicy_max','dvalue_max','within_bar')})
PY
