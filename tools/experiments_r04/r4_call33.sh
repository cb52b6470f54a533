#!/bin/bash
# bit-pattern lo codec (both paths) + fused stream: A/B, encoder suite on both paths (product library = LayerNorm path, experiment library = fused)
set -o pipefail
mkdir -p gpurun_out/r4c33
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c33/pytest_encoder_ln_path.txt 2>&1 || { tail -20 gpurun_out/r4c33/pytest_encoder_ln_path.txt; exit 1; }
tail -2 gpurun_out/r4c33/pytest_encoder_ln_path.txt
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
timeout -k 10 500 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c33/ab_fused.txt || { cat gpurun_out/r4c33/ab_fused.txt; exit 1; }
cat gpurun_out/r4c33/ab_fused.txt
KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c33/pytest_encoder_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c33/pytest_encoder_fused.txt
tail -4 gpurun_out/r4c33/pytest_encoder_fused.txt
