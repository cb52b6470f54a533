#!/bin/bash
# round 5, call 9: split search (exchange before the re-rank): single-process two-shard test, gloo world-2 tests with device shards, whole search suite
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c9; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_search.py tests/test_gpu_distributed.py tests/test_gpu_config5.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
