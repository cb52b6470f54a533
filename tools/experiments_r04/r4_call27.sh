#!/bin/bash
# fused stream: which projection path produces the non-finite values?  Forced tilings / store policies on 8 x 128 and 64 x 64 tokens, compared with the LayerNorm path
set -o pipefail
mkdir -p gpurun_out/r4c27
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so AB_SHAPES=8x128,64x64
for cfg in "KIRAG_AMD_PROJ_TILE=128" "KIRAG_AMD_PROJ_TILE=130" "KIRAG_AMD_PROJ_TILE=256" "KIRAG_AMD_PROJ_TILE=256 KIRAG_AMD_STORE_NT=1" "KIRAG_AMD_PROJ_TILE=128 KIRAG_AMD_STORE_NT=1"; do
  echo "== $cfg" >> gpurun_out/r4c27/forced.txt
  env $cfg timeout -k 10 200 python tools/ab_fused.py 2>&1 | grep -v amdgpu | tail -3 >> gpurun_out/r4c27/forced.txt
done
cat gpurun_out/r4c27/forced.txt
