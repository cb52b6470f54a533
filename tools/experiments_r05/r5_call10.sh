#!/bin/bash
# round 5, call 10: strong-scaling projection with the split search (per-rank work on one GPU), driver's --steps 20
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c10; mkdir -p $O
cd $R
timeout -k 10 900 python3 tools/scale_emulate.py 5000000 20 2>&1 | grep -v amdgpu.ids | tee $O/scale_emulate.txt
