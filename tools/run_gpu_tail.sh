#!/bin/bash
# per-launch fixed cost of the projection kernels: batches of 512 n x 32 tokens = exactly n (out-proj, FF2), 3 n (QKV), 4 n (FF1) 256x256 tiles per CU
mkdir -p gpurun_out
for n in 1 2 3 4 6; do
  B=$((512 * n))
  bash tools/shape_trace.sh $B 32 5 > gpurun_out/tail_shape_$n.txt 2>&1
  echo "== n=$n"; grep "k_proj\|shape" gpurun_out/tail_shape_$n.txt | cut -c1-140
done
