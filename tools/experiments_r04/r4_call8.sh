#!/bin/bash
# round 4, GPU call 8: do non-temporal loads of the once-read activation streams (h in FF2, q / k / v in attention) keep the residual stream in the Infinity Cache for LayerNorm?
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export KIRAG_AMD_LIB=$R/tools/bin/libkirag_exp.so
cd $R
timeout -k 10 300 python tools/ab_encoder.py KIRAG_AMD_NT_H=0,1 2>&1 | grep -v amdgpu > gpurun_out/r4c8_ab_nt_h.txt || exit 1
cat gpurun_out/r4c8_ab_nt_h.txt
KIRAG_AMD_NT_H=1 timeout -k 10 300 python tools/ab_encoder.py KIRAG_AMD_NT_QKV=0,1 2>&1 | grep -v amdgpu > gpurun_out/r4c8_ab_nt_qkv.txt || exit 1
cat gpurun_out/r4c8_ab_nt_qkv.txt
cd /tmp && export TMPDIR=/tmp
for tag in base both; do
  if [ $tag = both ]; then export KIRAG_AMD_NT_H=1 KIRAG_AMD_NT_QKV=1; else unset KIRAG_AMD_NT_H KIRAG_AMD_NT_QKV; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4c8_$tag -- python3 $R/tools/one_shape.py 1000 32 10 > /dev/null 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/r4c8_$tag/*/*kernel_stats.csv"))[-1]
print("== $tag")
for r in csv.DictReader(open(f)):
    if "kr::" in r["Name"] and float(r["Percentage"]) > 1.0:
        print("  %-60s calls %5s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
