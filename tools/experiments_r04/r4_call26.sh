#!/bin/bash
# Fused residual stream after the two fixes (bit_cast of a vector element; sign extension in the uniform-pointer helper): numerics + timing A/B, encoder suite on the fused path
set -o pipefail
mkdir -p gpurun_out/r4c26
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
KIRAG_AMD_SYNC_EACH=1 KIRAG_AMD_FUSED_LN=1 timeout -k 10 200 python tools/one_shape.py 32 128 1 > gpurun_out/r4c26/sync_each_32x128.txt 2>&1 || { grep -v amdgpu gpurun_out/r4c26/sync_each_32x128.txt | tail -5; exit 1; }
timeout -k 10 500 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c26/ab_fused.txt || { cat gpurun_out/r4c26/ab_fused.txt; exit 1; }
cat gpurun_out/r4c26/ab_fused.txt
KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c26/pytest_encoder_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c26/pytest_encoder_fused.txt
tail -12 gpurun_out/r4c26/pytest_encoder_fused.txt
