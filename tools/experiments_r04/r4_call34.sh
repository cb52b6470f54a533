#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c34
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c34/pytest_encoder_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c34/pytest_encoder_fused.txt
grep -n "G10 benign\|passed\|failed\|rc=" gpurun_out/r4c34/pytest_encoder_fused.txt | head
AB_SHAPES=1000x32,1024x128,125x32 timeout -k 10 500 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c34/ab_fused.txt || { cat gpurun_out/r4c34/ab_fused.txt; exit 1; }
cat gpurun_out/r4c34/ab_fused.txt
