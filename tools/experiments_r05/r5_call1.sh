#!/bin/bash
# round 5, call 1: seam microbenchmark + where the small-batch forward's time goes today
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c1; mkdir -p $O
cd $R
timeout -k 10 120 tools/bin/seam_bench 24 7 > $O/seam_bench.txt 2>&1
cat $O/seam_bench.txt
timeout -k 10 300 python3 tools/small_timeline.py > $O/small_wall.txt 2>&1
cat $O/small_wall.txt
cd /tmp && export TMPDIR=/tmp
for sh in "1 32" "1 256" "8 128"; do
  set -- $sh
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1_$2 -- python3 $R/tools/one_shape.py $1 $2 12 > /dev/null 2>&1
  python3 $R/tools/small_timeline.py $O/trace_$1_$2 | tee -a $O/small_trace.txt
done
