#!/bin/bash
# Fused residual stream, second run: (1) the LayerNorm path after the dispatch refactor (encoder parity suite, product library), (2) fused at sizes of the
# 128x128 loops, (3) fused on 256x256 tiles with a wait after every launch (which kernel faulted in call 23?), (4) timing A/B
set -o pipefail
mkdir -p gpurun_out/r4c24
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c24/pytest_encoder_baseline.txt 2>&1 || { tail -20 gpurun_out/r4c24/pytest_encoder_baseline.txt; exit 1; }
tail -2 gpurun_out/r4c24/pytest_encoder_baseline.txt
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
AB_SHAPES=125x32,32x128 timeout -k 10 300 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c24/ab_fused_small.txt || { cat gpurun_out/r4c24/ab_fused_small.txt; exit 1; }
cat gpurun_out/r4c24/ab_fused_small.txt
KIRAG_AMD_SYNC_EACH=1 KIRAG_AMD_PROJ_TILE=256 KIRAG_AMD_FUSED_LN=1 timeout -k 10 200 python tools/one_shape.py 8 128 1 > gpurun_out/r4c24/sync_each_big.txt 2>&1; rc=$?
grep -v amdgpu gpurun_out/r4c24/sync_each_big.txt | head -12; grep -v amdgpu gpurun_out/r4c24/sync_each_big.txt | tail -4
[ $rc -eq 0 ] || exit 1
AB_SHAPES=1000x32,1024x128 timeout -k 10 300 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c24/ab_fused_big.txt || { cat gpurun_out/r4c24/ab_fused_big.txt; exit 1; }
cat gpurun_out/r4c24/ab_fused_big.txt
