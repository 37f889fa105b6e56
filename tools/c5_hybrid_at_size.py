"""Scratch (GPU): one whole move at C5's per-GPU size (4096 games x 800 sims, 20x256) in pure f16x3 and in hybrid, under
the size-independent invariants of tests/test_gpu_at_size.py -- and the two must build the same trees, visit for visit
(3.3 million S1 evaluations: the hybrid mode's f16 pass + indexed layer-wise fall-back against the layer-wise f16x3
evaluation of every board).  python tools/c5_hybrid_at_size.py [G=4096] [sims=800] [blocks=20] [filters=256]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_gpu_at_size import _one_move_at_size

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 20
filters = int(sys.argv[4]) if len(sys.argv) > 4 else 256
t0 = time.time()
a = _one_move_at_size(G, sims, blocks, filters, seed=5, precision="f16x3")
t1 = time.time()
torch.cuda.empty_cache()
h = _one_move_at_size(G, sims, blocks, filters, seed=5, precision="hybrid")
t2 = time.time()
print("%d games x %d sims, %dx%d: one whole move in f16x3 %.1f s, in hybrid %.1f s; visits equal: %s, nchild equal: %s; depth %.2f"
      % (G, sims, blocks, filters, t1 - t0, t2 - t1, np.array_equal(a["visits"], h["visits"]),
         np.array_equal(a["nchild"], h["nchild"]), a["depth"]))
assert np.array_equal(a["visits"], h["visits"]) and np.array_equal(a["nchild"], h["nchild"])
