#!/bin/bash
# round 5, call 26: byte pre-scan appends its marked rows itself (no bitmap / compaction), counters zeroed by the prep kernels, 1024-thread select for small
# blocks, small calls work in the caller's buffers: search tests, A/B at 5M rows, hop at the surface
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c26; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_search.py tests/test_gpu_distributed.py tests/test_gpu_lifecycle.py tests/test_gpu_config5.py tests/test_gpu_surface.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 500 python3 tools/experiments_r05/byte_scan_check.py 5000000 gaussian > $O/byte_prescan_ab_5M.txt 2>&1 || { tail -20 $O/byte_prescan_ab_5M.txt; exit 1; }
cat $O/byte_prescan_ab_5M.txt
timeout -k 10 200 python3 tools/hop_surface.py > $O/hop_surface.txt 2>&1 || { tail -20 $O/hop_surface.txt; exit 1; }
head -5 $O/hop_surface.txt
timeout -k 10 300 python3 tests/soak_gpu.py 120 77 > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
tail -2 $O/soak.txt
