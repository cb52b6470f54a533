#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_encoder.py -q -x > gpurun_out/r4c13_pytest_enc.txt 2>&1; rc=$?; tail -3 gpurun_out/r4c13_pytest_enc.txt
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" gpurun_out/r4c13_pytest_enc.txt | head -20; exit $rc; fi
AB_SHAPES=1000x32,250x32,1024x128,128x512 timeout -k 10 300 python tools/ab_encoder.py KIRAG_AMD_TRICKLE=1,0 2>&1 | grep -v amdgpu > gpurun_out/r4c13_trickle.txt || exit 1
cat gpurun_out/r4c13_trickle.txt
