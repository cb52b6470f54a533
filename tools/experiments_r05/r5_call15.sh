#!/bin/bash
# round 5, call 15: whole GPU suite + smoke on the final library
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c15; mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
