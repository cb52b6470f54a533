#!/bin/bash
# Data-dependent MFMA power: coarse copies (corpus and queries) rounded to 1 / 2 / 3 fewer mantissa bits (bf16: 7 -> 6 / 5 / 4 explicit bits); the
# certificate uses the measured |x - c(x)|, so results stay exact and the cost shows up as re-ranked rows.  Experiment build, interleaved A/B.
set -o pipefail
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
timeout -k 10 500 python tools/ab_search.py KIRAG_AMD_TRIM_X+KIRAG_AMD_TRIM_Q=0,1,2,3 2>&1 | grep -v amdgpu > gpurun_out/r4c21_trim.txt || exit 1
cat gpurun_out/r4c21_trim.txt
