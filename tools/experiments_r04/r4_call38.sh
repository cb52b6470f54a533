#!/bin/bash
# fused stream: compile-time low-half switch + row scales prefetched for the next tile.  Suite on the fused path, A/B, kernel breakdown
set -o pipefail
mkdir -p gpurun_out/r4c38
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c38/pytest_encoder_fused.txt 2>&1 || { tail -20 gpurun_out/r4c38/pytest_encoder_fused.txt; exit 1; }
tail -1 gpurun_out/r4c38/pytest_encoder_fused.txt
AB_SHAPES=125x32,1000x32,1024x128,128x512 timeout -k 10 500 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c38/ab_fused.txt || { cat gpurun_out/r4c38/ab_fused.txt; exit 1; }
cat gpurun_out/r4c38/ab_fused.txt
rm -rf gpurun_out/r4c32; bash tools/experiments_r04/r4_call32.sh 2>&1 | grep -E "fused=|k_proj|k_ln16|row_stats|kernel time" | cut -c1-130
