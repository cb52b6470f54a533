#!/bin/bash
# round 5, call 40: candidate counters of consecutive queries 4 KiB apart (-DKR_CNT_PITCH=1024) against adjacent ones, search-only bench, alternating on one box
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c40; mkdir -p $O
cd $R
KIRAG_AMD_LIB=$R/tools/bin/libkirag_cntpitch.so timeout -k 10 300 python3 -m pytest tests/test_gpu_search.py -x -q -m gpu -k "small_exact or few_queries or multi_round or byte_prescan_small or three_rounds or k_200" > $O/tests_pitch.log 2>&1 || { tail -30 $O/tests_pitch.log; exit 1; }
tail -1 $O/tests_pitch.log
for rnd in 1 2 3; do
for v in adjacent pitch1024; do
  if [ $v = adjacent ]; then unset KIRAG_AMD_LIB; else export KIRAG_AMD_LIB=$R/tools/bin/libkirag_cntpitch.so; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-surface --no-encoder > $O/b_${v}_$rnd.json 2> $O/b_${v}_$rnd.err || { tail -5 $O/b_${v}_$rnd.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_${v}_$rnd.json')); print('$v round $rnd: %.3f ms per step (search only), coarse %.3f ms' % (d['ms_per_step'], d['roofline']['launch_ms']))" | tee -a $O/ab_cnt_pitch.txt
done
done
