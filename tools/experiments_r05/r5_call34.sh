#!/bin/bash
# round 5, call 34: 16-bit list threshold raised to kth16 - 2 eps16: byte tests, A/B at 5M rows, 150-s soak
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c34; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_search.py -x -q -m gpu -k "byte_prescan or split or pass2 or few_queries" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python3 tools/experiments_r05/byte_scan_check.py 5000000 > $O/byte_prescan_ab_5M.txt 2>&1 || { tail -20 $O/byte_prescan_ab_5M.txt; exit 1; }
grep "k  10\|k 100" $O/byte_prescan_ab_5M.txt | grep "near"
timeout -k 10 400 python3 tests/soak_gpu.py 150 4321 > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
grep "byte pre-scan" $O/soak.txt | tail -1; tail -1 $O/soak.txt
