#!/bin/bash
# round 5, call 18: headline step A/B again (LayerNorm decode without the non-finite check) + the overflow tests
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c18; mkdir -p $O
cd $R
for rnd in 1 2; do
for v in r04 new; do
  if [ $v = new ]; then unset KIRAG_AMD_LIB KIRAG_AMD_LIB_OLDER; else export KIRAG_AMD_LIB=$R/tools/bin/libkirag_$v.so KIRAG_AMD_LIB_OLDER=1; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-surface > $O/b_${v}_$rnd.json 2> $O/b_${v}_$rnd.err || { tail -5 $O/b_${v}_$rnd.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_${v}_$rnd.json')); print('$v round $rnd: %.2f ms per step, coarse %.3f ms, encode %.0f passages/s' % (d['ms_per_step'], d['roofline']['launch_ms'], d['encode']['passages_per_s']))" | tee -a $O/ab_step.txt
done
done
unset KIRAG_AMD_LIB KIRAG_AMD_LIB_OLDER
timeout -k 10 600 python3 -m pytest tests/test_gpu_lifecycle.py tests/test_gpu_encoder.py -x -q -m gpu -k "overflow or projection_paths or g2_full or cls_pooling" 2>&1 | tail -2
