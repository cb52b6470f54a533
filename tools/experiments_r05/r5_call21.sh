#!/bin/bash
# round 5, call 21: byte pre-scan of small query blocks — A/B against the 16-bit final round at 1M and 5M rows
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c21; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/experiments_r05/byte_scan_check.py 1000000 gaussian 2>&1 | tee $O/check_1M.txt
timeout -k 10 500 python3 tools/experiments_r05/byte_scan_check.py 5000000 2>&1 | tee $O/check_5M.txt
