#!/bin/bash
set -o pipefail
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so AB_SHAPES=1000x32,1024x128
AB_NOCHECK=1 timeout -k 10 300 python tools/ab_encoder.py KIRAG_AMD_SKIP_STORES_FROM=99,2,0 2>&1 | grep -v amdgpu > gpurun_out/r4c12_skip_stores.txt || exit 1
cat gpurun_out/r4c12_skip_stores.txt
