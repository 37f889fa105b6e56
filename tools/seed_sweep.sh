#!/bin/bash
# What `auto` decides for eight random-init C3 nets and what each run proves about it: bench.py --seed s for s = 0..7, the
# key fields of every line collected into gpurun_out/auto_decisions_c3_seeds.json (probe, chosen mode, rate, the 5120-position
# check against the fp32 oracle, the run-time guard's record, in-step roofline).   bash tools/seed_sweep.sh [seeds="0 1 .. 7"]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
SEEDS=${1:-"0 1 2 3 4 5 6 7"}
EXTRA=${EXTRA:-}        # e.g. EXTRA="--blocks 20 --filters 256" TAG=c5 for the same sweep at another configuration
TAG=${TAG:-c3}
mkdir -p gpurun_out/seed_sweep
for s in $SEEDS; do
  python bench.py --seed $s --steps 20 --warmup 5 --cpu-seconds 1 --strict-steps 0 $EXTRA > gpurun_out/seed_sweep/bench_${TAG}_seed$s.json 2> /dev/null
done
TAG=$TAG EXTRA="$EXTRA" python - <<'PY'
import glob, json, os
TAG, EXTRA = os.environ["TAG"], os.environ["EXTRA"]
rows = []
for f in sorted(glob.glob("gpurun_out/seed_sweep/bench_%s_seed*.json" % TAG)):
    d = json.load(open(f))
    c, p, r = d["config"], d["tower_error_vs_fp32"], d["roofline"]
    rows.append({"seed": int(f.split("seed")[-1].split(".")[0]), "probe": c["tower_precision_probe"], "mode": c["tower_precision"],
                 "why": c["tower_precision_why"][:80], "simulations_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                 "vs_fp32": {k: p[k] for k in ("positions", "dpolicy_max", "dvalue_max", "dvalue_p999", "positions_beyond_bar", "within_bar")},
                 "guard": c.get("tower_precision_guard"), "roofline_frac": r["frac"], "launch_ms": r["launch_ms"],
                 "s1_boards_evaluated_twice": (d.get("precision_modes") or {}).get("hybrid", {}).get("s1_boards_evaluated_twice"),
                 "distinct_root_positions": d["window"].get("distinct_root_positions"),
                 "trunk_in_graph_ms": {k: v["launch_ms"] for k, v in (r.get("step_fit", {}).get("trunk_in_step") or {}).items()},
                 "step_fit_ratio": r.get("step_fit", {}).get("ratio")})
    print(rows[-1]["seed"], rows[-1]["mode"], round(rows[-1]["simulations_per_s"]), rows[-1]["vs_fp32"]["dvalue_max"], rows[-1]["guard"])
json.dump({"what": "bench.py --seed s --steps 20 --warmup 5 %s (4096 games, 800 sims/move, random init; no extra flags = C3, 10x128), one line per seed" % EXTRA,
           "rows": rows}, open("gpurun_out/auto_decisions_%s_seeds.json" % TAG, "w"), indent=1)
PY
