#!/bin/bash
# round 5, call 14: k_pool with 8 tokens per step, packing kernels merged for small batches: encoder + surface suites, wall times, per-kernel breakdown
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c14; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu_surface.py tests/test_gpu_aligner.py tests/test_gpu_lifecycle.py -x -q -m gpu -k "not g10 and not checkpoint" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
REPS=30 timeout -k 10 200 python3 tools/small_timeline.py 2>&1 | grep " x " | tee $O/small_wall.txt
cd /tmp && export TMPDIR=/tmp
for sh in "1 32" "1 256" "8 128"; do
  set -- $sh
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1_$2 -- python3 $R/tools/one_shape.py $1 $2 12 > /dev/null 2>&1
  python3 $R/tools/trace_breakdown.py $O/trace_$1_$2 | tail -12 | tee -a $O/breakdown.txt
done
