#!/bin/bash
# k_rerank average duration under rocprofv3 for two builds of the library (bash tools/rerank_ab.sh libA libB)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dist in gaussian e5like; do
for lib in "$@"; do
  export KIRAG_AMD_LIB=$R/$lib
  tag=$(basename $lib .so)_$dist
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rr_$tag -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-encoder --corpus-dist $dist > /dev/null 2>&1
  echo "$dist $lib: $(grep -h 'k_rerank\|k_select' $R/gpurun_out/rr_$tag/*/*kernel_stats.csv | awk -F'",' '{split($1,a,"("); print a[1], $2, $4}' | tr '\n' ';')"
done
done
