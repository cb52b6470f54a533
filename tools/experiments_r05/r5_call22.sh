#!/bin/bash
# round 5, call 22: byte pre-scan — its GPU tests, the search test file, a 200-s soak with the path forced on at every size
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c22; mkdir -p $O
cd $R
true
true
timeout -k 10 420 python3 tests/soak_gpu.py 300 505 > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
tail -4 $O/soak.txt
