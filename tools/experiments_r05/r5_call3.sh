#!/bin/bash
# round 5, call 3: timing model of a 32-token forward, persistent vs launches, with weight streaming (go / no-go for the single-launch forward)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c3; mkdir -p $O
cd $R
timeout -k 10 200 tools/bin/seam_bench2 7 > $O/seam_bench2.txt 2>&1
cat $O/seam_bench2.txt
