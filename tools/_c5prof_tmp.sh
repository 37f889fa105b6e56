cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05; mkdir -p $O
rm -rf /tmp/st_c5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_c5 -- python3 $R/bench.py --no-cpu-baseline --parity-positions 0 --blocks 20 --filters 256 --steps 100 --warmup 20 --no-graph > $O/bench_c5_profiled.json 2> $O/bench_c5.err
find /tmp/st_c5 -name "*kernel_stats.csv" -exec cp {} $O/bench_c5_kernel_stats.csv \;
find /tmp/st_c5 -name "*domain_stats.csv" -exec cp {} $O/bench_c5_domain_stats.csv \;
ls -la $O/bench_c5*
cd $R
python bench.py --no-cpu-baseline --parity-positions 0 --blocks 20 --filters 256 --steps 40 --warmup 8 --no-graph --strict-steps 0 > gpurun_out/r05_bench_c5_nograph.json 2>/dev/null
python -c "
import json
for f in ('gpurun_out/r05_bench_c5_nograph.json','gpurun_out/prof_r05/bench_c5_profiled.json'):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['config']['tower_precision'], d['roofline']['step_fit']['sum_ms'], d['roofline']['step_fit']['ratio'])
    except Exception as e: print(f, 'ERR', e)
"
bash tools/scale_rehearsal.sh 8
