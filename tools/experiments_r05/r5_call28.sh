#!/bin/bash
# round 5, call 28: batch_retrieve keeps the query embeddings on the device: surface tests + hop at the surface
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c28; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_surface.py tests/test_gpu_config5.py tests/test_gpu_aligner.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python3 tools/hop_surface.py > $O/hop_surface.txt 2>&1 || { tail -20 $O/hop_surface.txt; exit 1; }
head -5 $O/hop_surface.txt
