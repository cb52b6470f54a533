#!/bin/bash
# final state of the round: full GPU suite + smoke + bench on the product library; the encoder suite once more on the fused path (experiment library)
set -o pipefail
mkdir -p gpurun_out/r4c39
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4c39/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/r4c39/pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/r4c39/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r4c39/bench.json 2> gpurun_out/r4c39/bench.err || { tail -5 gpurun_out/r4c39/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c39/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "q/s", round(d["value"]), "k_coarse ms", d["roofline"]["launch_ms"], "frac", round(d["roofline"]["frac"], 4), "passages/s", round(d["encode"]["passages_per_s"]), "enc frac", round(d["encode"]["frac_of_mfma_peak"], 4))
PY
KIRAG_AMD_LIB=tools/bin/libkirag_exp.so KIRAG_AMD_FUSED_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -q > gpurun_out/r4c39/pytest_encoder_fused.txt 2>&1; tail -1 gpurun_out/r4c39/pytest_encoder_fused.txt
