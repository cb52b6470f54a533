#!/bin/bash
# round 5, call 27: bitmap marking back (fire-and-forget), block-aggregated self-cleaning compaction: byte tests + A/B at 5M rows
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c27; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_search.py -x -q -m gpu -k "byte_prescan or pass2 or near_duplicate or fine_pass or few_queries" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 500 python3 tools/experiments_r05/byte_scan_check.py 5000000 gaussian > $O/byte_prescan_ab_5M.txt 2>&1 || { tail -20 $O/byte_prescan_ab_5M.txt; exit 1; }
cat $O/byte_prescan_ab_5M.txt
