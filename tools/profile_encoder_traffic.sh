#!/bin/bash
# HBM-side traffic of the encoder kernels per shape (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, program directly after `--`):
#   bash tools/profile_encoder_traffic.sh <tag> -> gpurun_out/<tag>/enc_traffic_<B>_<S>.json (tools/pmc_encoder_traffic.py)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in "1024 128" "128 512" "1000 32"; do
  set -- $shape
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/enc_${ctr}_$1_$2 -- python3 $R/tools/one_shape.py $1 $2 3 > /dev/null 2> $OUT/enc_${ctr}_$1_$2.err || echo "pass $ctr $1 $2 failed"
  done
  python3 $R/tools/pmc_encoder_traffic.py $OUT $1 $2 3 > $OUT/enc_traffic_$1_$2.json
  echo "encoder traffic $1 x $2 done"
done
