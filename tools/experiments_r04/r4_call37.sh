#!/bin/bash
# after the v_ashr_pk_u8_i32 work-around: stored rows consistent with the device statistics and identical across tilings?  encoder suite on both paths, A/B
set -o pipefail
mkdir -p gpurun_out/r4c37
bash tools/experiments_r04/r4_call36.sh > /dev/null 2>&1; cp gpurun_out/r4c36/consistency.txt gpurun_out/r4c37/consistency.txt; cat gpurun_out/r4c37/consistency.txt
unset KIRAG_AMD_LIB KIRAG_AMD_SYNC_EACH KIRAG_AMD_DBG_ROWS KIRAG_AMD_FUSED_LN
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/r4c37/pytest_encoder_ln_path.txt 2>&1 || { tail -20 gpurun_out/r4c37/pytest_encoder_ln_path.txt; exit 1; }
tail -1 gpurun_out/r4c37/pytest_encoder_ln_path.txt
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so
KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q -s > gpurun_out/r4c37/pytest_encoder_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c37/pytest_encoder_fused.txt
grep -n "G10 .* f16 lo=1\|passed\|failed\|rc=" gpurun_out/r4c37/pytest_encoder_fused.txt | head
timeout -k 10 500 python tools/ab_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r4c37/ab_fused.txt || { cat gpurun_out/r4c37/ab_fused.txt; exit 1; }
cat gpurun_out/r4c37/ab_fused.txt
