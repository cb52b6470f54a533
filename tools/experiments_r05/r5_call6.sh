#!/bin/bash
# round 5, call 6: pipelined search_knn (single index + row-sharded over gloo world 2), hardened finish_deferred, bench line with latency + surface blocks
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c6; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_search.py tests/test_gpu_distributed.py tests/test_gpu_surface.py tests/test_gpu_capi_c.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"])
print("latency", json.dumps(d.get("latency")))
print("surface", json.dumps(d.get("surface")))
print("encode", d["encode"]["passages_per_s"], d["encode"]["frac_of_mfma_peak"])
PY
