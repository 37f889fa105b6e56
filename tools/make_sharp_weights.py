"""Scratch (GPU): a SHARP weight file (peaked policy, values spread over (-1, 1), unit-gain layers: what a trained net looks
like to the precision modes) for bench rehearsals -- `oracle/tower_oracle.calibrated_weights` on real self-play positions.
python tools/make_sharp_weights.py out.npz [blocks=6] [filters=64] [seed=7]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chessrl_amd.model import ChessModel
from oracle import tower_oracle
from tests.util import encode_prefixes, selfplay_position_prefixes

out = sys.argv[1]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
filters = int(sys.argv[3]) if len(sys.argv) > 3 else 64
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
prefixes, info = selfplay_position_prefixes(512)
_, planes = encode_prefixes(ChessModel(blocks=2, filters=64, precision="f16"), prefixes)
w = tower_oracle.calibrated_weights(blocks, filters, planes[:512], seed=seed)
np.savez(out, **w)
print("wrote", out, blocks, filters, info)
