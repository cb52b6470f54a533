#!/bin/bash
# One parameterised entry point for this round's GPU-box runs (no numbered per-call scratch scripts).
#   gpurun --timeout N -- 'bash tools/gpu_round.sh <step> [args]'
# Every step writes under gpurun_out/r06/; the summaries that are judged are copied into profiles/r06/ by hand afterwards.
set -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06; mkdir -p $OUT
step=$1; shift
case "$step" in
  tests)        # [pytest args]  e.g. tests/test_gpu_encoder.py -k packed
    timeout -k 10 1100 python -m pytest "$@" -x -q -m gpu 2>&1 | tee $OUT/tests_$(date +%H%M%S).log ;;
  suite)        # the whole GPU suite + smoke, as the driver runs it
    timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/suite.log 2>&1; rc=$?; tail -30 $OUT/suite.log
    [ $rc -eq 0 ] && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $OUT/smoke.log ;;
  feed)         # [passages]  tools/feed_bench.py
    timeout -k 10 1000 python tools/feed_bench.py "${1:-100000}" 2>&1 | grep -v Warning | tee $OUT/feed_bench.txt ;;
  bench)        # [bench.py args]
    timeout -k 10 1000 python bench.py "$@" 2>&1 | tee $OUT/bench_$(date +%H%M%S).json ;;
  py)           # <script> [args]: any tool under tools/
    name=$(basename "$1" .py); timeout -k 10 1000 python "$@" 2>&1 | tee $OUT/${name}_$(date +%H%M%S).txt ;;
  *) echo "unknown step $step"; exit 2 ;;
esac
