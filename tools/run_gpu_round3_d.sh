set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r3_gputest5.log 2>&1; echo exit=$? >> gpurun_out/r3_gputest5.log
grep -E "passed|failed|exit=" gpurun_out/r3_gputest5.log | tail -3
grep -E "\[G10 " gpurun_out/r3_gputest5.log | grep "f16 lo=1"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_d.json 2> gpurun_out/r3_bench_d.err; python -c "
import json; b=json.load(open('gpurun_out/r3_bench_d.json')); print(round(b['value']), 'q/s', round(b['ms_per_step'],2), 'ms', round(b['encode']['passages_per_s']), 'p/s', round(b['roofline']['frac'],3))"
python tools/scale_emulate.py > gpurun_out/r3_scale_emulate.txt 2>&1; cat gpurun_out/r3_scale_emulate.txt
