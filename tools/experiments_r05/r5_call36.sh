#!/bin/bash
# round 5, call 36: HBM-side traffic of the one-query search's kernels (two separate --pmc passes, kernel-trace only)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c36; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/FETCH_SIZE -- python3 $R/tools/experiments_r05/byte_scan_profile.py > $O/fetch.log 2>&1 || { tail -20 $O/fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/WRITE_SIZE -- python3 $R/tools/experiments_r05/byte_scan_profile.py > $O/write.log 2>&1 || { tail -20 $O/write.log; exit 1; }
cd $R
python3 tools/experiments_r05/byte_scan_traffic.py $O 5000000 | tee $O/byte_scan_traffic.json
find $O -name "*.csv" -size +2M -delete
