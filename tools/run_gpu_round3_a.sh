set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s > gpurun_out/r3_gputest2.log 2>&1; echo exit=$? >> gpurun_out/r3_gputest2.log
grep -E "passed|failed|exit=" gpurun_out/r3_gputest2.log | tail -3
for m in "f16 1" "f16 0" "bf16 1" "bf16 0"; do set -- $m; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --encoder-dtype $1 --residual-lo $2 > gpurun_out/r3_bench_$1_lo$2.json 2> gpurun_out/r3_bench_$1_lo$2.err; python -c "
import json,sys; b=json.load(open('gpurun_out/r3_bench_$1_lo$2.json')); print('$1 lo$2', round(b['value']), 'q/s', round(b['ms_per_step'],2), 'ms', round(b['encode']['passages_per_s']), 'p/s', round(b['roofline']['frac'],3))"; done
