#!/bin/bash
# refresh of the secondary measurements with the round's final kernels: KiRAG-loop hop latencies (config 5), config 4's corpus size search-only, full GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r4c16_pytest.txt 2>&1; tail -3 gpurun_out/r4c16_pytest.txt
timeout -k 10 300 python tools/loop_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r4c16_loop_bench.txt || exit 1
cat gpurun_out/r4c16_loop_bench.txt
timeout -k 10 400 python bench.py --total-rows 21015324 --no-encoder --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > gpurun_out/r4c16_bench_21M.json || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r4c16_bench_21M.json')); print('21M search-only', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['launch_ms'], d['search_stats'])"
