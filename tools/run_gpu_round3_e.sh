set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_aligner.py tests/test_gpu_config5.py -m gpu -q -x > gpurun_out/r3_gputest6.log 2>&1; echo exit=$? >> gpurun_out/r3_gputest6.log
grep -E "passed|failed|exit=" gpurun_out/r3_gputest6.log | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_e.json 2> gpurun_out/r3_bench_e.err; python -c "
import json; b=json.load(open('gpurun_out/r3_bench_e.json')); print(round(b['value']), 'q/s', round(b['ms_per_step'],2), 'ms', round(b['encode']['passages_per_s']), 'p/s', round(b['roofline']['frac'],3))"
bash tools/shape_trace.sh 1000 32 5 > gpurun_out/r3_shape_1000_32.txt 2>&1; grep -E "shape|attn" gpurun_out/r3_shape_1000_32.txt
bash tools/profile_encoder_traffic.sh r03
