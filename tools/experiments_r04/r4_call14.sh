#!/bin/bash
# same-box process-level A/B: previous commit's library (pointer stores in the epilogue) vs this one with KIRAG_AMD_TRICKLE=0 (buffer stores, no trickling) and =1
set -o pipefail
export AB_SHAPES=1000x32,1024x128
for r in 1 2; do
  for cfg in "prev tools/bin/libkirag_prev.so 1" "buf kirag_amd/libkirag_amd.so 0" "trickle kirag_amd/libkirag_amd.so 1"; do
    set -- $cfg
    echo "== round $r: $1"
    KIRAG_AMD_LIB=$2 KIRAG_AMD_TRICKLE=$3 timeout -k 10 200 python tools/ab_encoder.py KIRAG_AMD_UNUSED=0 2>&1 | grep -v amdgpu | sed 's/KIRAG_AMD_UNUSED=0: //; s/  outputs identical: True//' || exit 1
  done
done
