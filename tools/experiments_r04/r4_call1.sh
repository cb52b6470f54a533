#!/bin/bash
# round 4, GPU call 1: parity of the buffer_load-lds staging + zero-spill kernels, A/B against round 3's library, k_coarse epilogue variants
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4c1_pytest.txt 2>&1; rc=$?
tail -3 gpurun_out/r4c1_pytest.txt
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; exit $rc; fi
timeout -k 10 500 bash tools/ab_libs.sh tools/bin/libkirag_r03.so kirag_amd/libkirag_amd.so 2 > gpurun_out/r4c1_ab_encoder.txt 2>&1 || exit 1
echo "ab encoder done"
: > gpurun_out/r4c1_bench_ab.txt
for i in 1 2; do
  for lib in tools/bin/libkirag_r03.so kirag_amd/libkirag_amd.so; do
    echo "== $lib" >> gpurun_out/r4c1_bench_ab.txt
    KIRAG_AMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 >> gpurun_out/r4c1_bench_ab.txt || exit 1
  done
done
echo "bench ab done"
KIRAG_AMD_LIB=tools/bin/libkirag_exp.so AB_NOCHECK=1 timeout -k 10 500 python tools/ab_search.py KIRAG_AMD_EPIV=0,1,2,3 > gpurun_out/r4c1_epiv.txt 2>&1 || exit 1
cat gpurun_out/r4c1_epiv.txt | tail -5
