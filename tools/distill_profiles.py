"""Turn the raw rocprofv3 output of tools/profile_pass.sh (gpurun_out/prof_<round>/) into the tracked
evidence: profiles/<round>/*.csv (kernel stats as rocprofv3 wrote them), profiles/<round>/pmc_summary.json
and the PMC table bench.py reads (profiles/pmc_traffic.json).  python tools/distill_profiles.py [r03]"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)
DST = os.path.join(ROOT, "profiles", ROUND)
os.makedirs(DST, exist_ok=True)


def per_kernel(path, last=None):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name].append(float(r["Counter_Value"]))
    return {k: (sum(v[-last:] if last else v) / len(v[-last:] if last else v), len(v)) for k, v in acc.items()}


passes, summary = [], {}
shapes = {"trunk128": "4096 boards, 10 blocks x 128 filters", "trunk256": "4096 boards, 20 blocks x 256 filters",
          "trunk64": "512 boards, 6 blocks x 64 filters", "trunk128x3": "4096 boards, 10 blocks x 128 filters",
          "trunk256x3": "4096 boards, 20 blocks x 256 filters"}
for tag, shape in shapes.items():
    if not os.path.exists(os.path.join(SRC, tag + "_FETCH_SIZE.csv")):
        continue
    f = per_kernel(os.path.join(SRC, tag + "_FETCH_SIZE.csv"))
    w = per_kernel(os.path.join(SRC, tag + "_WRITE_SIZE.csv"))
    for k in f:
        if "k_trunk" not in k:
            continue
        short = k.split("::")[-1]                             # the name crl_trunk_kernel_name reports
        passes.append({"kernel": short, "shape": shape, "fetch_size_kb": round(f[k][0], 1),
                       "write_size_kb": round(w[k][0], 1),
                       "source": "profiles/%s/pmc_summary.json (rocprofv3 --pmc, tools/trunk_once.py, %d launches)" % (ROUND, f[k][1])})
        summary[short + " @ " + shape] = {"FETCH_SIZE_KB": f[k][0], "WRITE_SIZE_KB": w[k][0]}
# the layer-wise split-precision trunk of 256 filters (csrc/tower_layer.hpp) is 1 + 1 + 2 x blocks launches per forward:
# traffic per FORWARD = the sum over its kernels, keyed by the name crl_trunk_kernel_name reports for that path
LAYER_NAME = "k_layer_conv<8, 1|2|3, 0, 4> (+ k_layer_conv<4, 0, 0, 4>, k_layer_expand<1, 0, 4>)"
for tag in ("trunk256x3",):
    pf, pw = os.path.join(SRC, tag + "_FETCH_SIZE.csv"), os.path.join(SRC, tag + "_WRITE_SIZE.csv")
    if not os.path.exists(pf):
        continue
    f, w = per_kernel(pf), per_kernel(pw)
    lay = [k for k in f if "k_layer_" in k]
    if lay:
        n_fwd = 3                                             # tools/trunk_once.py launches three forwards
        fetch = sum(f[k][0] * f[k][1] for k in lay) / n_fwd
        write = sum(w[k][0] * w[k][1] for k in lay) / n_fwd
        passes.append({"kernel": LAYER_NAME, "shape": shapes[tag], "fetch_size_kb": round(fetch, 1), "write_size_kb": round(write, 1),
                       "source": "profiles/%s/pmc_summary.json (rocprofv3 --pmc, tools/trunk_once.py, sum over the %d launches of one "
                                 "forward, mean of %d forwards)" % (ROUND, sum(f[k][1] for k in lay) // n_fwd, n_fwd)})
        summary[LAYER_NAME + " @ " + shapes[tag]] = {
            "FETCH_SIZE_KB_per_forward": fetch, "WRITE_SIZE_KB_per_forward": write,
            "per_kernel_mean_per_launch": {k.split("::")[-1]: {"FETCH_SIZE_KB": f[k][0], "WRITE_SIZE_KB": w[k][0], "launches": f[k][1]} for k in lay}}

# SQ counters (one pass of four + GRBM_GUI_ACTIVE): per kernel, mean per launch; derived: MFMA busy share of the SIMD-cycles
# of the launch (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): MI355X_MICROARCH.md), LDS bank-conflict
# cycles per LDS instruction
def sq_table(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, ctr in acc.items():
        if "k_trunk" not in k and "k_layer" not in k:
            continue
        e = {c: sum(v) / len(v) for c, v in ctr.items()}
        e["launches"] = len(next(iter(ctr.values())))
        if e.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_share_of_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if e.get("SQ_INSTS_LDS"):
            e["lds_bank_conflict_cycles_per_lds_instruction"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_INSTS_LDS"]
        out[k.split("::")[-1]] = e
    return out


sq = {}
for tag, what in (("trunk128", "C3 f16 (what auto keeps for soft nets)"), ("trunk128x3", "C3 split precision (S2 of hybrid / f16x3)"),
                  ("trunk128idx", "C3 hybrid fall-back launch, every 13th board listed"),
                  ("trunk256x3", "C5 split precision, layer-wise"), ("trunk256", "C5 f16")):
    path = os.path.join(SRC, tag + "_SQ.csv")
    if os.path.exists(path):
        sq[what] = sq_table(path)
if sq:
    summary["SQ counters (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE, "
            "tools/trunk_once.py, 4096 boards; mean per launch)"] = sq

for tag, fmt in (("tree", "legal priors"), ("treefull", "full policies")):
    if not os.path.exists(os.path.join(SRC, tag + "_FETCH_SIZE.csv")):
        continue
    tf = per_kernel(os.path.join(SRC, tag + "_FETCH_SIZE.csv"), last=50)
    tw = per_kernel(os.path.join(SRC, tag + "_WRITE_SIZE.csv"), last=50)
    tree = {k.split("::")[-1]: {"FETCH_SIZE_KB": tf[k][0], "WRITE_SIZE_KB": tw[k][0], "launches": tf[k][1]}
            for k in tf if "k_select_expand" in k or "k_reply" in k}
    shape_file = os.path.join(SRC, tag + "_shape.json")
    tree_shape = json.load(open(shape_file)) if os.path.exists(shape_file) else {}
    summary["search kernels, 4096 games, bit planes, %s (mean of the last 50 launches)" % fmt] = dict(tree, tree=tree_shape)
    passes.append({"kernel": "k_select_expand + k_reply", "shape": "4096 games, bit planes, %s" % fmt,
                   "fetch_size_kb": round(sum(v["FETCH_SIZE_KB"] for v in tree.values()), 1),
                   "write_size_kb": round(sum(v["WRITE_SIZE_KB"] for v in tree.values()), 1),
                   "source": "profiles/%s/pmc_summary.json (tools/tree_once.py 4096 400 1, %s, last 50 launches; %s)"
                             % (ROUND, fmt, json.dumps(tree_shape))})
json.dump(summary, open(os.path.join(DST, "pmc_summary.json"), "w"), indent=1)
table = {"what": "HBM-side traffic per launch from rocprofv3 PMC passes (one counter per pass; FETCH_SIZE and "
                 "WRITE_SIZE in KB as rocprofv3 reports them; bench.py applies the gfx950 x2 wide-read correction to "
                 "FETCH_SIZE). Keyed by the kernel name rocprofv3 prints (= what crl_trunk_kernel_name reports) and "
                 "the launch shape; bench.py emits traffic only for an exact match.",
         "passes": passes}
json.dump(table, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
for n in ("c3", "c5", "c5g", "c2"):
    for kind in ("kernel_stats.csv", "domain_stats.csv", "profiled.json"):
        src = os.path.join(SRC, "bench_%s_%s" % (n, kind))
        if os.path.exists(src):
            shutil.copy(src, os.path.join(DST, "bench_%s_%s" % (n, kind)))
print(json.dumps(summary, indent=1))
