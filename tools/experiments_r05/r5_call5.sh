#!/bin/bash
# round 5, call 5: whole GPU suite on the cleaned-up library (no KR_EXPERIMENT code, 64x64 skinny tile, no LN tail) + small-batch wall times
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c5; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
REPS=30 timeout -k 10 200 python3 tools/small_timeline.py 2>&1 | grep " x " | tee $O/small_wall.txt
