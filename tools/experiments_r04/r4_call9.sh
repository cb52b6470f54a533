#!/bin/bash
# round 4, GPU call 9: cache-policy / grid switches of the LayerNorm and of the out-projection's activation operand (-DKR_EXPERIMENT build), interleaved A/B in one process each
set -o pipefail
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so AB_SHAPES=1000x32,1024x128,128x512
OUT=gpurun_out/r4c9_policies.txt; : > $OUT
for spec in KIRAG_AMD_NT_CTX=0,1 KIRAG_AMD_LN_POL=0,1,7 KIRAG_AMD_LN_POL=0,6,5,3 KIRAG_AMD_LN_GRID=4,8,16,64; do
  echo "== $spec" >> $OUT
  timeout -k 10 300 python tools/ab_encoder.py $spec 2>&1 | grep -v amdgpu >> $OUT || exit 1
done
cat $OUT
