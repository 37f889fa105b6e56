#!/bin/bash
# What `auto` decides for eight random-init C3 nets and what each run proves about it: bench.py --seed s for s = 0..7, the
# key fields of every line collected into gpurun_out/auto_decisions_c3_seeds.json (probe, chosen mode, rate, the 5120-position
# check against the fp32 oracle, the run-time guard's record, in-step roofline).   bash tools/seed_sweep.sh [seeds="0 1 .. 7"]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
SEEDS=${1:-"0 1 2 3 4 5 6 7"}
mkdir -p gpurun_out/seed_sweep
for s in $SEEDS; do
  python bench.py --seed $s --steps 20 --warmup 5 --cpu-seconds 1 --strict-steps 0 > gpurun_out/seed_sweep/bench_c3_seed$s.json 2> /dev/null
done
python - <<'PY'
import glob, json
rows = []
for f in sorted(glob.glob("gpurun_out/seed_sweep/bench_c3_seed*.json")):
    d = json.load(open(f))
    c, p, r = d["config"], d["tower_error_vs_fp32"], d["roofline"]
    rows.append({"seed": int(f.split("seed")[-1].split(".")[0]), "probe": c["tower_precision_probe"], "mode": c["tower_precision"],
                 "why": c["tower_precision_why"][:80], "simulations_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                 "vs_fp32": {k: p[k] for k in ("positions", "dpolicy_max", "dvalue_max", "dvalue_p999", "positions_beyond_bar", "within_bar")},
                 "guard": c.get("tower_precision_guard"), "roofline_frac": r["frac"], "launch_ms": r["launch_ms"],
                 "step_fit_ratio": r.get("step_fit", {}).get("ratio")})
    print(rows[-1]["seed"], rows[-1]["mode"], round(rows[-1]["simulations_per_s"]), rows[-1]["vs_fp32"]["dvalue_max"], rows[-1]["guard"])
json.dump({"what": "bench.py --seed s --steps 20 --warmup 5 at C3 (4096 games, 800 sims/move, 10x128 random init), one line per seed",
           "rows": rows}, open("gpurun_out/auto_decisions_c3_seeds.json", "w"), indent=1)
PY
