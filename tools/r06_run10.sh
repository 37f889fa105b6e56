#!/bin/bash
# round 6, GPU run 10: parity at size for the round's kernel changes: one whole C5 move in f16x3 and in hybrid (3.3 M S1 evaluations
# through the f16 pass + the reworked indexed fall-back) must build the same trees; the 512-game soak against the CPU oracle
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python tools/c5_hybrid_at_size.py > $O/c5_hybrid_at_size.log 2>&1; echo "c5 at size rc $?"; tail -2 $O/c5_hybrid_at_size.log
timeout 900 python tools/c5_hybrid_at_size.py 2048 200 10 256 > $O/ref_net_hybrid_at_size.log 2>&1; echo "10x256 at size rc $?"; tail -2 $O/ref_net_hybrid_at_size.log
timeout 900 python tools/soak_parity.py 512 8 128 > $O/soak_512_games.log 2>&1; echo "soak rc $?"; tail -3 $O/soak_512_games.log
