"""Scratch (GPU): search-kernel phase times of one library build (CRL_LIB_PATH selects it)."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "400", "--warmup", "100"],
                     capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
t = d["roofline_tree"]
print(os.environ.get("CRL_LIB_PATH", "product"), "sims/s %.0f" % d["value"], {k: round(v, 4) for k, v in t["phase_ms"].items()}, "depth %.2f" % t["mean_depth"])
