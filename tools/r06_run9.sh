#!/bin/bash
# round 6, GPU run 9: what auto decides and what each mode sustains over eight random-init C3 nets and four C5 nets, on the
# round-6 workload (noisy opening moves: games that differ)
cd $GRAFT_REPO_ROOT
EXTRA="--gph-seconds 0" TAG=c3 timeout 1500 bash tools/seed_sweep.sh "0 1 2 3 4 5 6 7"
EXTRA="--gph-seconds 0 --blocks 20 --filters 256" TAG=c5 timeout 1800 bash tools/seed_sweep.sh "0 1 2 3"
