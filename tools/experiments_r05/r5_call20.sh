#!/bin/bash
# round 5, call 20: final tree — whole GPU suite, smoke(), and the plain bench line for profiles/r05/bench_plain.json
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c20; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
cat $O/bench_plain.json
