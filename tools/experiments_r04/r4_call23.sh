#!/bin/bash
# Fused residual stream (no LayerNorm kernels; KIRAG_AMD_FUSED_LN=1), first measurement: times against the LayerNorm path + the encoder parity suite on it
set -o pipefail
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
mkdir -p gpurun_out/r4c23
timeout -k 10 400 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c23/ab_fused.txt || { cat gpurun_out/r4c23/ab_fused.txt; exit 1; }
cat gpurun_out/r4c23/ab_fused.txt
KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c23/pytest_encoder_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c23/pytest_encoder_fused.txt
tail -15 gpurun_out/r4c23/pytest_encoder_fused.txt
