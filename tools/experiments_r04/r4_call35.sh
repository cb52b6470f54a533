#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c35
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q -s -k "g10_full_size_long_sequences and benign" > gpurun_out/r4c35/g10_benign_ln_path.txt 2>&1
grep -n "G10 benign\|passed\|failed" gpurun_out/r4c35/g10_benign_ln_path.txt
