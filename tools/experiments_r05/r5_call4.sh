#!/bin/bash
# round 5, call 4b: skinny prologue order A/B (early weights: none = default / 2 tiles) against round 4 library, single-compare wait; LN tail off in all
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c4b; mkdir -p $O
cd $R
export KIRAG_AMD_LN_TAIL=0 REPS=30
for rnd in 1 2; do
for v in r04 e2 new; do
  if [ $v = new ]; then unset KIRAG_AMD_LIB; else export KIRAG_AMD_LIB=$R/tools/bin/libkirag_$v.so; fi
  echo "== $v (round $rnd)" | tee -a $O/ab.txt
  timeout -k 10 200 python3 tools/small_timeline.py 2>&1 | grep " x " | tee -a $O/ab.txt
done
done
