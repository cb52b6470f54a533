#!/bin/bash
# round 5, call 7: FETCH_SIZE of k_coarse, round 3's package (global_load_lds staging) against this round's library (buffer_load ... lds staging),
# ALTERNATING on one box, three passes each, + one TCC hit / miss pass each
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5c7; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # tag, which (r03tree | new), counters...
  tag=$1; which=$2; shift 2
  BENCH=$R/bench.py
  if [ "$which" = "r03tree" ]; then BENCH=$R/tools/bin/r03/bench.py; fi
  extra=""; if [ "$which" = "new" ]; then extra="--no-latency --no-surface"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -- python3 $BENCH --steps 3 --warmup 1 --no-encoder --no-cpu-baseline $extra > $OUT/$tag.json 2> $OUT/$tag.err || { echo "pass $tag failed"; tail -5 $OUT/$tag.err; return 1; }
  python3 - "$@" <<PY
import csv, glob, json, sys
names = sys.argv[1:]
tot = {n: 0.0 for n in names}; nd = 0
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") in tot and "k_coarse" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd += 1
d = json.load(open("$OUT/$tag.json"))
parts = []
for n in names:
    if n == "FETCH_SIZE": parts.append("FETCH_SIZE x2 = %.2f GB per scan" % (tot[n] / 4 * 1024 * 2 / 1e9))
    else: parts.append("%s = %.3e per scan" % (n, tot[n] / 4))
print("$tag: " + ", ".join(parts) + ", coarse %.3f ms per scan (%d counter rows)" % (d["roofline"]["launch_ms"], nd))
PY
}
for i in 1 2 3; do
  pass r03_$i r03tree FETCH_SIZE | tee -a $OUT/summary.txt
  pass new_$i new FETCH_SIZE | tee -a $OUT/summary.txt
done
pass r03_tcc r03tree TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum | tee -a $OUT/summary.txt
pass new_tcc new TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum | tee -a $OUT/summary.txt
