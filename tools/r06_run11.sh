#!/bin/bash
# round 6, GPU run 11: the run-time check of hybrid's reply rule: its test, the tower tests, a whole C3 round in hybrid
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_measure.py -x -q > $O/test_rule.log 2>&1; echo "rule test rc $?"; tail -15 $O/test_rule.log
