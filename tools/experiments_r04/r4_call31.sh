#!/bin/bash
# fused stream after the store-data wait: tilings bit-identical?  then the A/B and the encoder suite on the fused path
set -o pipefail
mkdir -p gpurun_out/r4c31
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
for cfg in "KIRAG_AMD_PROJ_TILE=256" "KIRAG_AMD_PROJ_TILE=256 KIRAG_AMD_STORE_NT=1"; do
  echo "== $cfg" >> gpurun_out/r4c31/forced.txt
  env $cfg AB_SHAPES=8x128,64x64 timeout -k 10 200 python tools/ab_fused.py 2>&1 | grep -v amdgpu | tail -3 >> gpurun_out/r4c31/forced.txt
done
cat gpurun_out/r4c31/forced.txt
timeout -k 10 500 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c31/ab_fused.txt || { cat gpurun_out/r4c31/ab_fused.txt; exit 1; }
cat gpurun_out/r4c31/ab_fused.txt
KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c31/pytest_encoder_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c31/pytest_encoder_fused.txt
tail -12 gpurun_out/r4c31/pytest_encoder_fused.txt
