#!/bin/bash
# round 4, GPU call 5: soak of the round's kernels (buffer_load staging, pending ring) + the round's profile set
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 460 python tests/soak_gpu.py 400 141 > gpurun_out/r4c5_soak.txt 2>&1; rc=$?
tail -2 gpurun_out/r4c5_soak.txt
if [ $rc -ne 0 ]; then echo "soak failed rc=$rc"; exit $rc; fi
bash tools/profile_round.sh r04 > gpurun_out/r4c5_profile.log 2>&1 || { tail -20 gpurun_out/r4c5_profile.log; exit 1; }
tail -3 gpurun_out/r4c5_profile.log
