#!/bin/bash
# round 4, GPU call 2: full GPU suite with the pending-ring / deferred sharded search / C host, vendor GEMM reference, corpus-from-cache coarse experiment, scale emulation
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r4c2_pytest.txt 2>&1; rc=$?
tail -3 gpurun_out/r4c2_pytest.txt
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc (continuing with the measurements)"; fi
timeout -k 10 300 python tools/vendor_gemm_ref.py > gpurun_out/r4c2_vendor_gemm.txt 2>&1 || exit 1
echo "vendor gemm done"
# coarse scan with the corpus served from the Infinity Cache (120k rows = 245 MB bf16, 4 x 1024 queries: the corpus is scanned 4 times per call) vs from HBM (5M rows)
for args in "--total-rows 120000 --queries 4096" "--total-rows 5000000 --queries 1000"; do
  echo "== $args" >> gpurun_out/r4c2_coarse_cache.txt
  timeout -k 10 300 python bench.py --no-encoder --no-cpu-baseline --sync-search --steps 10 $args 2>/dev/null | tail -1 >> gpurun_out/r4c2_coarse_cache.txt || exit 1
done
echo "coarse cache experiment done"
timeout -k 10 500 python tools/scale_emulate.py 5000000 20 > gpurun_out/r4c2_scale_emulate.txt 2>&1 || exit 1
tail -4 gpurun_out/r4c2_scale_emulate.txt
