#!/bin/bash
# round 5, call 8: new GPU tests (overflow in the last LayerNorm, checkpoint_check through the HIP encoder), attention kernel choice for one or two long
# sequences, shape traces of the reference's long-passage shapes, feed rate re-measured
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c8; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_lifecycle.py tests/test_gpu_encoder.py -x -q -m gpu -k "overflow or checkpoint_check or g10" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
cat > /tmp/attn_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from kirag_amd import bench_support as BS
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
for B, S in ((1, 256), (2, 256), (1, 512), (2, 512), (4, 256), (8, 256)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    res = {}
    for rnd in range(3):
        for mode in ("dma", "lds"):
            if mode == "lds": os.environ["KIRAG_AMD_ATTN_LDS"] = "1"
            else: os.environ.pop("KIRAG_AMD_ATTN_LDS", None)
            for _ in range(3): o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize(); res.setdefault(mode, []).append((time.perf_counter() - t0) / 30 * 1e3)
            res[mode + "_out"] = o.clone()
    print(f"{B} x {S}: ring kernel (default) {np.median(res['dma']):.3f} ms, register-staged kernel {np.median(res['lds']):.3f} ms, identical {torch.equal(res['dma_out'], res['lds_out'])}", flush=True)
PY
timeout -k 10 300 python3 /tmp/attn_ab.py 2>&1 | grep " x " | tee $O/attn_small_b.txt
bash tools/shape_trace.sh 128 512 6 | tee $O/shape_128_512.txt
bash tools/shape_trace.sh 256 256 6 | tee $O/shape_256_256.txt
timeout -k 10 600 python3 tools/feed_bench.py 100000 2>&1 | grep "^\[" | tee $O/feed_bench_100k.txt
