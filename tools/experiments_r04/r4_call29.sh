#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c29
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so KIRAG_AMD_SYNC_EACH=1 KIRAG_AMD_DBG_ROWS=1024 KIRAG_AMD_FUSED_LN=1
KIRAG_AMD_PROJ_TILE=256 timeout -k 10 200 python tools/one_shape.py 8 128 1 2>&1 | grep -v amdgpu > gpurun_out/r4c29/dump_256.txt
grep -n "after proj RES" -A 30 gpurun_out/r4c29/dump_256.txt | head -45
