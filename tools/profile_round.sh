#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/profile_round.sh r01'): kernel-trace stats of the default bench, kernel-trace of the search-only
# bench, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the search-only bench.  Summaries -> gpurun_out/<tag>/ (copy into profiles/).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_search -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encoder > $OUT/bench_search_under_rocprof.json 2> $OUT/kt_search.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-encoder --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-encoder --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 $R/tools/pmc_summary.py $OUT > $OUT/search_pmc_fetch_write.json
python3 $R/tools/pmc_summary.py $OUT --traffic 4 > $OUT/traffic.json
echo "profile_round done: $OUT"
