#!/usr/bin/env python3
"""CLS pooling (bge): last layer on the CLS rows only (default) vs all rows (KIRAG_AMD_CLS_FULL=1 at creation), two encoders interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kirag_amd import bench_support as BS
dev = torch.device("cuda:0")
encs = {}
for name, v in (("cls rows", "0"), ("all rows", "1")):
    os.environ["KIRAG_AMD_CLS_FULL"] = v
    encs[name] = BS.make_hip_encoder(dev)
os.environ.pop("KIRAG_AMD_CLS_FULL")
for B, S in ((1000, 32), (1024, 128), (128, 512), (8, 128), (2, 256)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    res = {k: [] for k in encs}; outs = {}
    for rnd in range(5):
        for k, enc in encs.items():
            for _ in range(2):
                o = enc.forward(ids, mask, 1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 5 if B * S > 20000 else 20
            for _ in range(reps):
                o = enc.forward(ids, mask, 1)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / reps * 1e3); outs[k] = o
    same = torch.equal(outs["cls rows"], outs["all rows"])
    print(f"{B} x {S} (CLS pooling): " + "  ".join(f"{k}: median {np.median(v):.3f} ms" for k, v in res.items()) + f"  bit-identical: {same}", flush=True)
