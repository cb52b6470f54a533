#!/bin/bash
# LayerNorm with two rows' loads in flight per wave (prefetch distance 2; 154 instead of 122 VGPRs: 3 instead of 4 waves per SIMD), experiment build, interleaved A/B
set -o pipefail
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so AB_SHAPES=1000x32,1024x128
timeout -k 10 300 python tools/ab_encoder.py KIRAG_AMD_LN_POL=7,15 2>&1 | grep -v amdgpu > gpurun_out/r4c20_ln_prefetch2.txt || exit 1
for g in 8 16; do
  echo "== KIRAG_AMD_LN_GRID=$g" >> gpurun_out/r4c20_ln_prefetch2.txt
  KIRAG_AMD_LN_GRID=$g timeout -k 10 300 python tools/ab_encoder.py KIRAG_AMD_LN_POL=7,15 2>&1 | grep -v amdgpu >> gpurun_out/r4c20_ln_prefetch2.txt || exit 1
done
cat gpurun_out/r4c20_ln_prefetch2.txt
