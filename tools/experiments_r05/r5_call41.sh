#!/bin/bash
# round 5, call 41: final library (search_knn with two blocks in flight) — whole GPU suite, smoke, plain bench line
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c41; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
python3 -c "
import json; d=json.load(open('$O/bench_plain.json')); print(d['ms_per_step'], d['roofline']['frac']); print(d['surface']); print(json.dumps(d['latency']['kirag_hop_nq1'])[:400])"
