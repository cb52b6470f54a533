#!/bin/bash
# encoder parity tests + per-kernel times of the long-sequence shapes (bash tools/run_gpu_attn.sh <tag>)
TAG=${1:-a}
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q > gpurun_out/attn_${TAG}_tests.log 2>&1; rc=$?
tail -n 4 gpurun_out/attn_${TAG}_tests.log
[ $rc -eq 0 ] || exit $rc
for shape in "128 512" "256 256" "1000 32" "1024 128"; do
  set -- $shape
  bash tools/shape_trace.sh $1 $2 5 > gpurun_out/attn_${TAG}_shape_$1_$2.txt 2>&1
  grep "shape\|attn" gpurun_out/attn_${TAG}_shape_$1_$2.txt
done
