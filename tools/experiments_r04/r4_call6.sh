#!/bin/bash
# round 4, GPU call 6: does running the search of step i on a SUBSET of the CUs under the encode of step i + 1 (other CUs, second stream) beat the serial step?
# (-DKR_EXPERIMENT build: KIRAG_AMD_ENC_CUS / KIRAG_AMD_IDX_CUS size the persistent grids; bench.py --search-stream puts the search on a side stream)
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r4c6_cu_split.txt; : > $OUT
run() {  # label, enc cus, idx cus, extra flags
  echo "== $1" >> $OUT
  KIRAG_AMD_LIB=tools/bin/libkirag_exp.so KIRAG_AMD_ENC_CUS=$2 KIRAG_AMD_IDX_CUS=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 $4 2>/dev/null | tail -1 >> $OUT || exit 1
}
run "serial, all CUs" 256 256 ""
run "two streams, all CUs each (round 2's experiment)" 256 256 "--search-stream"
run "two streams, encoder 192 / search 64" 192 64 "--search-stream"
run "two streams, encoder 176 / search 80" 176 80 "--search-stream"
run "two streams, encoder 160 / search 96" 160 96 "--search-stream"
run "two streams, encoder 128 / search 128" 128 128 "--search-stream"
run "serial, all CUs (again)" 256 256 ""
python - <<'PY'
import json
for ln in open("gpurun_out/r4c6_cu_split.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln); print("   ms/step %.2f  queries/s %.0f" % (d["ms_per_step"], d["value"]))
PY
