#!/bin/bash
# HBM-side traffic of k_coarse (FETCH_SIZE, rocprofv3 --pmc) for several builds of the library, ALTERNATING on one box (is the 1.17x -> 1.28x ->
# 1.38x drift of roofline.traffic a property of the kernel or of the box?).   gpurun -- 'bash tools/traffic_ab.sh <tag> <passes> name=lib.so [name=lib.so ...]'
# ("name=" alone = the tree's own library).  Older libraries are loaded through KIRAG_AMD_LIB + KIRAG_AMD_LIB_OLDER=1 (kirag_amd/_lib.py).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; PASSES=$2; shift 2
OUT=$R/gpurun_out/r06/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $PASSES); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ -n "$lib" ]; then export KIRAG_AMD_LIB=$R/$lib KIRAG_AMD_LIB_OLDER=1; else unset KIRAG_AMD_LIB KIRAG_AMD_LIB_OLDER; fi
    tag=${name}_$i
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-encoder --no-cpu-baseline --no-latency --no-surface > $OUT/$tag.json 2> $OUT/$tag.err || { echo "pass $tag failed"; tail -5 $OUT/$tag.err; exit 1; }
    python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, json
tot = 0.0; nd = 0
for f in glob.glob("$OUT/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE" and "k_coarse" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); nd += 1
d = json.load(open("$OUT/$tag.json"))
print("$tag: FETCH_SIZE x2 = %.2f GB per scan (%.2fx of 10.24 GB), coarse %.3f ms per scan under the counters (%d counter rows)" % (tot / 4 * 1024 * 2 / 1e9, tot / 4 * 1024 * 2 / 1.024e10, d["roofline"]["launch_ms"], nd))
PY
    rm -rf $OUT/$tag
  done
done
