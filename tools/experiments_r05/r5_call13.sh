#!/bin/bash
# round 5, call 13: row count as a kernel argument for small launches (KIRAG_AMD_TFIX): parity, then one-process A/B
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c13; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu_surface.py -x -q -m gpu -k "not g10 and not checkpoint" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
cat > /tmp/tfix_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from kirag_amd import bench_support as BS
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
for B, S in ((1, 32), (1, 256), (2, 256), (4, 64), (8, 128), (125, 32)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1, ragged=(B > 100))
    res = {}
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ["KIRAG_AMD_TFIX"] = mode
            for _ in range(3): o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize(); res.setdefault(mode, []).append((time.perf_counter() - t0) / 30 * 1e3)
            res[mode + "_out"] = o.clone()
    print(f"{B} x {S}: row count from device memory {np.median(res['0']):.3f} ms, as an argument {np.median(res['1']):.3f} ms, identical {torch.equal(res['0_out'], res['1_out'])}", flush=True)
PY
timeout -k 10 300 python3 /tmp/tfix_ab.py 2>&1 | grep " x " | tee $O/tfix_ab.txt
