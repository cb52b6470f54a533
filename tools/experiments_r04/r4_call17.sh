#!/bin/bash
# does putting the search of step i on a second stream (full-size grids; it can only fill launch gaps / tails) pay?  alternating runs on one box
set -o pipefail
OUT=gpurun_out/r4c17_two_streams.txt; : > $OUT
for r in 1 2 3; do
  for flag in "" "--search-stream"; do
    echo "== round $r: ${flag:-serial}" >> $OUT
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 $flag 2>/dev/null | tail -1 >> $OUT || exit 1
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r4c17_two_streams.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln); print("   ms/step %.3f  queries/s %.0f" % (d["ms_per_step"], d["value"]))
PY
