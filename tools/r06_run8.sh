#!/bin/bash
# round 6, GPU run 8: is the persistent tower's -3 % (run 2) a property of the kernel or of the box?  (round 5's box: +0.1 %)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 ./tools/ubench/tower_persist 4096 20 5 2>&1 | grep -v "staggered" | tee $O/tower_persist_$(date +%H%M%S).log
timeout 300 python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -2
