"""Scratch (GPU): what ChessModel(precision="auto") measures and decides -- the f16-vs-f16x3 distance on its
probe positions -- for random-init nets of several seeds and, optionally, weight files.
python tools/probe_decisions.py [blocks=10] [filters=128] [seeds=0,1,2,3] [weight files ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chessrl_amd.model import ChessModel
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
filters = int(sys.argv[2]) if len(sys.argv) > 2 else 128
seeds = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3").split(",")]
for seed in seeds:
    m = ChessModel(blocks=blocks, filters=filters, seed=seed)
    print(json.dumps({"net": "%dx%d seed %d" % (blocks, filters, seed), **m.precision_probe}), flush=True)
for path in sys.argv[4:]:
    m = ChessModel(weights=path)
    print(json.dumps({"net": os.path.basename(path), **m.precision_probe}), flush=True)
