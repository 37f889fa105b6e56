#!/bin/bash
# Scratch (GPU): the -m gpu parity tests while a second process keeps the GPU busy
# (tools/trunk_stability_probe.py disturb): results that depend on timing show up as failures.
#   bash tools/disturbed_suite.sh [pytest files ...]     (default: rules, search, self-play)
cd $GRAFT_REPO_ROOT
python tools/trunk_stability_probe.py ${DISTURB:-disturb} 100000 > /dev/null 2>&1 &
D=$!
FILES=${@:-tests/test_gpu_rules.py tests/test_gpu_search.py tests/test_gpu_selfplay.py}
python -m pytest $FILES -q -x -m gpu 2>&1 | tail -5
RC=${PIPESTATUS[0]}
kill $D; wait $D 2>/dev/null
exit $RC
