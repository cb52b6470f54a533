set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r3_gputest4.log 2>&1; echo exit=$? >> gpurun_out/r3_gputest4.log
grep -E "passed|failed|exit=" gpurun_out/r3_gputest4.log | tail -3
bash tools/shape_trace.sh 128 512 5 > gpurun_out/r3_shape_128_512_dma.txt 2>&1; grep -E "shape|attn" gpurun_out/r3_shape_128_512_dma.txt
bash tools/shape_trace.sh 256 256 5 > gpurun_out/r3_shape_256_256_dma.txt 2>&1; grep -E "shape|attn" gpurun_out/r3_shape_256_256_dma.txt
python bench.py --steps 10 --warmup 3 > gpurun_out/r3_bench_c.json 2> gpurun_out/r3_bench_c.err; python -c "
import json; b=json.load(open('gpurun_out/r3_bench_c.json')); print(round(b['value']), 'q/s', round(b['ms_per_step'],2), 'ms', round(b['encode']['passages_per_s']), 'p/s', round(b['roofline']['frac'],3), b['cpu_baseline']['value'], b['cpu_baseline']['search_points'])"
