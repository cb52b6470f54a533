#!/bin/bash
# round 4, GPU call 7: FETCH_SIZE of k_coarse on ONE box for round 3's library, this round's, and this round's with default-policy corpus loads (EPIV=4)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4c7; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # tag, lib ("r03tree" = round 3's whole package, tools/bin/r03), epiv
  BENCH=$R/bench.py
  if [ "$2" = "r03tree" ]; then unset KIRAG_AMD_LIB; BENCH=$R/tools/bin/r03/bench.py; else export KIRAG_AMD_LIB=$2; fi
  export KIRAG_AMD_EPIV=$3
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$1 -- python3 $BENCH --steps 3 --warmup 1 --no-encoder --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err || { echo "pass $1 failed"; tail -5 $OUT/$1.err; exit 1; }
  python3 - <<PY
import csv, glob, json
tot = 0.0; n = 0
for f in glob.glob("$OUT/$1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE" and "k_coarse" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
d = json.load(open("$OUT/$1.json"))
print("$1: k_coarse FETCH_SIZE x2 = %.2f GB per scan (%d dispatches / 4 scans), coarse %.3f ms per scan" % (tot / 4 * 1024 * 2 / 1e9, n, d["roofline"]["launch_ms"]))
PY
}
pass r03 r03tree 0
pass r04 $R/kirag_amd/libkirag_amd.so 0
pass r04_no_nt $R/tools/bin/libkirag_exp.so 4
pass r03_again r03tree 0
