#!/bin/bash
# Diagnostic build of the library with in-kernel cycle stamps (gemm_nt.hpp, -DKR_STAMP) -> tools/bin/libkirag_amd_stamp.so (never the product .so).
set -e
cd "$(dirname "$0")/../kirag_amd/csrc"
mkdir -p ../../tools/bin/stamp
for f in capi_common search encoder; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -ffp-contract=off -DKR_STAMP -c $f.hip -o ../../tools/bin/stamp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libkirag_amd_stamp.so ../../tools/bin/stamp/*.o
echo built tools/bin/libkirag_amd_stamp.so
