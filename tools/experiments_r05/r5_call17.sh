#!/bin/bash
# round 5, call 17: the headline step, round 4's library against this round's, alternating on one box (+ loop_bench for profiles/r05/loop_bench.txt)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c17; mkdir -p $O
cd $R
for rnd in 1 2 3; do
for v in r04 new; do
  if [ $v = new ]; then unset KIRAG_AMD_LIB KIRAG_AMD_LIB_OLDER; else export KIRAG_AMD_LIB=$R/tools/bin/libkirag_$v.so KIRAG_AMD_LIB_OLDER=1; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-surface > $O/b_${v}_$rnd.json 2> $O/b_${v}_$rnd.err || { tail -5 $O/b_${v}_$rnd.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_${v}_$rnd.json')); print('$v round $rnd: %.2f ms per step, coarse %.3f ms, encode %.0f passages/s' % (d['ms_per_step'], d['roofline']['launch_ms'], d['encode']['passages_per_s']))" | tee -a $O/ab_step.txt
done
done
unset KIRAG_AMD_LIB KIRAG_AMD_LIB_OLDER
timeout -k 10 400 python3 tools/loop_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/loop_bench.txt
