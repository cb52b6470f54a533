#!/bin/bash
# round 5, call 12: FOUR K-tiles per barrier (ring of 16) against two: parity, then A/B (same box, two rounds)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c12; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "projection_paths or g1_tiny or g2_full or long_and_many or batch_invariance or cls_pooling" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
export REPS=30
for rnd in 1 2; do
for v in grp2 new; do
  if [ $v = new ]; then unset KIRAG_AMD_LIB; else export KIRAG_AMD_LIB=$R/tools/bin/libkirag_$v.so; fi
  echo "== $v (round $rnd)" | tee -a $O/ab.txt
  timeout -k 10 200 python3 tools/small_timeline.py 2>&1 | grep " x " | tee -a $O/ab.txt
done
done
