set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
python bench.py --steps 5 --warmup 2 --cpu-full-corpus > gpurun_out/r03/bench_cpu_full_corpus.json 2> gpurun_out/r03/bench_cpu_full_corpus.err; python -c "
import json; b=json.load(open('gpurun_out/r03/bench_cpu_full_corpus.json')); c=b['cpu_baseline']; print('cpu 5M', c['search_only_qps'], c['search_runs_s'], c['value'])
json.dump(c, open('gpurun_out/r03/cpu_baseline_5M.json','w'), indent=1)"
python tools/loop_bench.py > gpurun_out/r03/loop_bench.txt 2>&1; cat gpurun_out/r03/loop_bench.txt | tail -3
python tools/feed_bench.py 200000 > gpurun_out/r03/feed_bench_200k.txt 2>&1; tail -12 gpurun_out/r03/feed_bench_200k.txt
timeout -k 10 400 python tests/soak_gpu.py 240 71 > gpurun_out/r03/soak_240s_seed71.txt 2>&1; tail -4 gpurun_out/r03/soak_240s_seed71.txt
