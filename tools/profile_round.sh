#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/profile_round.sh r02'): kernel-trace stats of the default bench, kernel-trace of the search-only
# bench, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the search-only bench.  Summaries -> gpurun_out/<tag>/ (copy into profiles/).
# The program after `--` is python3 itself (never env / bash -c: the profiler's preloaded library has initialised the GPU).
set -e
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "plain bench done" 
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-surface --no-entry-point > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
echo "kernel trace (full step) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_search -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encoder > $OUT/bench_search_under_rocprof.json 2> $OUT/kt_search.err
echo "kernel trace (search only) done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-encoder --no-cpu-baseline --no-latency --no-surface > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-encoder --no-cpu-baseline --no-latency --no-surface > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write done"
# MFMA-pipe utilisation (north_star): counters in their own passes, program directly after `--`
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-surface --no-entry-point > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err || echo "pmc mfma pass failed"
echo "pmc mfma done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-surface --no-entry-point > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || echo "pmc sq pass failed"
echo "pmc sq done"
# HBM traffic of the encoder kernels (k_attn_lds, k_ln16, k_proj): FETCH / WRITE passes of the full step
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_enc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-surface --no-entry-point > $OUT/pmc_fetch_enc.json 2> $OUT/pmc_fetch_enc.err || echo "pmc fetch enc failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_enc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-surface --no-entry-point > $OUT/pmc_write_enc.json 2> $OUT/pmc_write_enc.err || echo "pmc write enc failed"
echo "pmc encoder traffic done"
python3 $R/tools/pmc_mfma.py $OUT/pmc_mfma k_coarse=4.1945e13 > $OUT/mfma_busy.json || true
python3 $R/tools/pmc_mfma.py $OUT/pmc_sq > $OUT/sq_wave_breakdown.json || true
python3 $R/tools/pmc_clock.py $OUT/pmc_mfma $OUT/pmc_mfma.json > $OUT/clock.json || true
python3 $R/tools/pmc_summary.py $OUT > $OUT/search_pmc_fetch_write.json
python3 $R/tools/pmc_summary.py $OUT --traffic 4 > $OUT/traffic.json
python3 $R/tools/trace_breakdown.py $OUT/kt > $OUT/encoder_forward_breakdown.txt
python3 $R/bench.py --steps 10 --warmup 2 --corpus-dist e5like --no-cpu-baseline --no-latency --no-surface --no-entry-point > $OUT/bench_e5like.json 2> $OUT/bench_e5like.err
python3 $R/bench.py --steps 10 --warmup 2 --total-rows 1000000 --no-cpu-baseline --no-latency --no-surface --no-entry-point > $OUT/bench_config2_1M.json 2> $OUT/bench_1M.err
# raw rocprofv3 output is large (the copy back from the GPU box stops at 64 MiB): keep the per-kernel statistics of the two kernel traces, drop the rest
for t in kt kt_search; do
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/$( [ $t = kt ] && echo bench_kernel_stats.csv || echo bench_search_only_kernel_stats.csv )
done
rm -rf $OUT/kt $OUT/kt_search $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_sq $OUT/pmc_fetch_enc $OUT/pmc_write_enc
du -sh $OUT
echo "profile_round done: $OUT"
