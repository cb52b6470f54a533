#!/bin/bash
# soak of the final library (300 s, seed 151), and of the fused residual stream in the experiment library (200 s, seed 152: the encoder cases check that the
# four projection loops agree bit for bit and that a sequence's embedding does not depend on its batch — on the fused path too)
set -o pipefail
mkdir -p gpurun_out/r4c40
timeout -k 10 360 python tests/soak_gpu.py 300 151 > gpurun_out/r4c40/soak_300s_seed151_final_library.txt 2>&1 || { tail -5 gpurun_out/r4c40/soak_300s_seed151_final_library.txt; exit 1; }
tail -2 gpurun_out/r4c40/soak_300s_seed151_final_library.txt
KIRAG_AMD_LIB=tools/bin/libkirag_exp.so KIRAG_AMD_FUSED_LN=1 timeout -k 10 260 python tests/soak_gpu.py 200 152 > gpurun_out/r4c40/soak_200s_seed152_fused.txt 2>&1 || { tail -5 gpurun_out/r4c40/soak_200s_seed152_fused.txt; exit 1; }
tail -2 gpurun_out/r4c40/soak_200s_seed152_fused.txt
