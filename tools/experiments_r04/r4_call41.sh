#!/bin/bash
# LayerNorm blocks per CU once more: the bit-pattern codec took k_ln16 from 210 to 142 VGPRs (3 waves per SIMD instead of 2)
set -o pipefail
mkdir -p gpurun_out/r4c41
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so AB_SHAPES=1000x32,1024x128
timeout -k 10 400 python tools/ab_encoder.py KIRAG_AMD_LN_GRID=8,12,16,24 2>&1 | grep -v amdgpu > gpurun_out/r4c41/ln_grid.txt || exit 1
cat gpurun_out/r4c41/ln_grid.txt
