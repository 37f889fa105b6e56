"""Scratch (GPU): whole self-play games (noise on, drawn ahead; refill; rolling rounds; compaction of the
last round's tail) against the CPU oracle, more of them than the test suite plays.
python tools/soak_parity.py [games=512] [sims=8] [round_size=128]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chessrl_amd.selfplay import SelfPlayRunner, game_color
from oracle.fakenet import FakeNet
from tests.util import oracle_games_parallel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rs = int(sys.argv[3]) if len(sys.argv) > 3 else 128
seed = 31
net = FakeNet(seed=77, prior_shift=30)
t0 = time.time()
run = SelfPlayRunner(net.to("cuda:0"), n_parallel=128, sims=sims, seed=seed, noise=True, total_games=n, round_size=rs)
recs = {}
run.run_rolling((n + rs - 1) // rs, on_round=lambda r, rr: recs.update({x.game_id: x for x in rr}))
print("GPU: %d games in %d rolling rounds of %d in %.1fs, final batch %d slots" % (len(recs), (n + rs - 1) // rs, rs, time.time() - t0, run.G))
t0 = time.time()
hist = oracle_games_parallel(dict(seed=77, prior_shift=30), seed, sims, [(g, game_color(seed, g)) for g in sorted(recs)],
                             workers=min(64, (os.cpu_count() or 2) - 1))
bad = 0
for gid in sorted(recs):
    h = hist[gid]
    ok = recs[gid].get_history()["moves"] == h["moves"] and recs[gid].result == h["result"]
    bad += not ok
    if not ok:
        print("MISMATCH game", gid)
print("oracle (%d workers): %.1fs; %d games, %d plies total, mismatches: %d" % (
    min(64, (os.cpu_count() or 2) - 1), time.time() - t0, len(recs), sum(len(r) for r in recs.values()), bad))
sys.exit(1 if bad or len(recs) != n else 0)
