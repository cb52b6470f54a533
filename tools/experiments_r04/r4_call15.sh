#!/bin/bash
# folded subtraction in the attention step (C operand of the S^T MFMAs = -mref) + k_attn_lds at two blocks per CU: encoder suite, then same-box A/B against the previous kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_encoder.py -q > gpurun_out/r4c15_pytest_enc.txt 2>&1; rc=$?; tail -3 gpurun_out/r4c15_pytest_enc.txt
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" gpurun_out/r4c15_pytest_enc.txt | head -20; fi
export AB_SHAPES=1000x32,1024x128,256x256,128x512
for r in 1 2; do
  for cfg in "prev tools/bin/libkirag_prev.so" "fold kirag_amd/libkirag_amd.so"; do
    set -- $cfg
    echo "== round $r: $1"
    KIRAG_AMD_LIB=$2 timeout -k 10 200 python tools/ab_encoder.py KIRAG_AMD_UNUSED=0 2>&1 | grep -v amdgpu | sed 's/KIRAG_AMD_UNUSED=0: //; s/  outputs identical: True//' || exit 1
  done
done
