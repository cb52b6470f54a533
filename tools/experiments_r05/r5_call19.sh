#!/bin/bash
# round 5, call 19: module-surface weight sync (cached walk) test + surface suite, bench line with the one-at-a-time module-surface latency
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c19; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_surface.py tests/test_gpu_aligner.py tests/test_gpu_config5.py -x -q -m gpu -s > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
grep "sync with the cached walk" $O/tests.log; tail -2 $O/tests.log
timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-surface > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(json.dumps(d['latency'], indent=1))"
