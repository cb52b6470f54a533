#!/bin/bash
# round 4, GPU call 10: GPU suite on the new defaults (FF2 nt loads, LayerNorm policies), then three more switches (A/B, experiment build)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r4c10_pytest.txt 2>&1; tail -3 gpurun_out/r4c10_pytest.txt
export KIRAG_AMD_LIB=tools/bin/libkirag_exp.so AB_SHAPES=1000x32,1024x128
OUT=gpurun_out/r4c10_switches.txt; : > $OUT
for spec in KIRAG_AMD_QKV_NT=1,0 KIRAG_AMD_PATCH_W=8,4,16 KIRAG_AMD_LN_POL=7,0; do
  echo "== $spec" >> $OUT
  timeout -k 10 300 python tools/ab_encoder.py $spec 2>&1 | grep -v amdgpu >> $OUT || exit 1
done
cat $OUT
unset KIRAG_AMD_LIB
python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4c10_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4c10_bench.json')); print('bench', d['ms_per_step'], d['roofline']['frac'], d['encode']['passages_per_s'], d['encode']['frac_of_mfma_peak'])"
