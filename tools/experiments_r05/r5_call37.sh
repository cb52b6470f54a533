#!/bin/bash
# round 5, call 37: byte round for 384-d rows (three-slot ring): byte tests + a 150-s soak, seed 384
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c37; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_search.py -x -q -m gpu -k "byte_prescan or split" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python3 tests/soak_gpu.py 150 384 > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
grep "byte pre-scan" $O/soak.txt | tail -1; tail -1 $O/soak.txt
