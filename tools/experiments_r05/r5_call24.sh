#!/bin/bash
# round 5, call 24: byte pre-scan with 4096-candidate buffers (two 16-bit rounds instead of four) + status records written by one kernel: search / distributed /
# lifecycle GPU tests, then the A/B table at 5M rows
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c24; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_search.py tests/test_gpu_distributed.py tests/test_gpu_lifecycle.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 500 python3 tools/experiments_r05/byte_scan_check.py 5000000 > $O/byte_prescan_ab_5M.txt 2>&1 || { tail -20 $O/byte_prescan_ab_5M.txt; exit 1; }
cat $O/byte_prescan_ab_5M.txt
