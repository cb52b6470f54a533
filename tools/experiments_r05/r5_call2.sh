#!/bin/bash
# round 5, call 2: LayerNorm tail + 64x64 skinny tile + early weight staging: parity, then latency A/B
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c2; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "projection_paths or layernorm_tail or g1_tiny or g2_full or long_and_many or batch_invariance or cls_pooling" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python3 tools/small_timeline.py > $O/small_wall_tail.txt 2>&1; cat $O/small_wall_tail.txt
KIRAG_AMD_LN_TAIL=0 timeout -k 10 300 python3 tools/small_timeline.py > $O/small_wall_notail.txt 2>&1; cat $O/small_wall_notail.txt
cd /tmp && export TMPDIR=/tmp
for sh in "1 32" "1 256" "8 128"; do
  set -- $sh
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1_$2 -- python3 $R/tools/one_shape.py $1 $2 12 > /dev/null 2>&1
  python3 $R/tools/trace_breakdown.py $O/trace_$1_$2 | tail -12 | tee -a $O/breakdown.txt
done
