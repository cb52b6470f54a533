#!/bin/bash
# round 5, call 23: byte pre-scan — A/B table at 5M rows, per-kernel times of the one-query search, the plain bench line (hop latency, headline unchanged)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c23; mkdir -p $O
cd $R
timeout -k 10 500 python3 tools/experiments_r05/byte_scan_check.py 5000000 > $O/byte_prescan_ab_5M.txt 2>&1 || { tail -20 $O/byte_prescan_ab_5M.txt; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_hop_search -o hop -- python3 $R/tools/experiments_r05/byte_scan_profile.py > $O/prof_hop_search.log 2>&1 || { tail -20 $O/prof_hop_search.log; exit 1; }
cd $R
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
python3 -c "
import json; d=json.load(open('$O/bench_plain.json')); print(d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d['latency'])[:1500])"
