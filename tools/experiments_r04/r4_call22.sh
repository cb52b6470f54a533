#!/bin/bash
# Small-batch forwards (the KiRAG loop's chain queries 2 x 256 tokens, and 1 x 32): kernel time against wall time per forward (is the rest launch gaps?),
# plus un-profiled wall time per forward with the knobs now read once per forward
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c22; mkdir -p $OUT
for shape in "2 256" "1 32" "8 128"; do
  tag=$(echo $shape | tr ' ' x)
  rocprofv3 --kernel-trace -d $OUT/kt_$tag -o t --output-format csv -- python3 $R/tools/one_shape.py $shape 12 > $OUT/log_$tag.txt 2>&1 || exit 1
  echo "== $shape" >> $OUT/small_forward_breakdown.txt
  python3 $R/tools/trace_breakdown.py $OUT/kt_$tag >> $OUT/small_forward_breakdown.txt || exit 1
  rm -rf $OUT/kt_$tag
done
cd $R
python3 - >> $OUT/small_forward_breakdown.txt <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
from kirag_amd import bench_support as BS
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
for B, S in ((2, 256), (1, 32), (8, 128), (4, 64)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    for _ in range(5): enc.forward(ids, mask, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): enc.forward(ids, mask, 0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ts = []
    for _ in range(20):
        a = time.perf_counter(); enc.forward(ids, mask, 0); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter(); ts.append((b - a, c - a))
    print(f"{B} x {S}: back to back {(t1 - t0) / 50 * 1e3:.3f} ms per forward; single: host enqueue {sorted(t[0] for t in ts)[10] * 1e3:.3f} ms, to completion {sorted(t[1] for t in ts)[10] * 1e3:.3f} ms")
PY
cat $OUT/small_forward_breakdown.txt
