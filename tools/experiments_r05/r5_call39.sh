#!/bin/bash
# round 5, call 39: final library — whole GPU suite, smoke, kernel trace of the one-query search, hop at the surface, plain bench line, 600-s soak
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c39; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_hop_search -o hop -- python3 $R/tools/experiments_r05/byte_scan_profile.py > $O/prof_hop_search.log 2>&1 || { tail -20 $O/prof_hop_search.log; exit 1; }
cd $R
timeout -k 10 200 python3 tools/hop_surface.py > $O/hop_surface.txt 2>&1 || { tail -20 $O/hop_surface.txt; exit 1; }
head -5 $O/hop_surface.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
python3 -c "
import json; d=json.load(open('$O/bench_plain.json')); print(d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d['latency']['kirag_hop_nq1'])[:700])"
timeout -k 10 800 python3 tests/soak_gpu.py 600 3031 > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
grep "byte pre-scan" $O/soak.txt | tail -1; tail -1 $O/soak.txt
