#!/usr/bin/env python3
"""A/B of the fused residual stream (KIRAG_AMD_FUSED_LN, read at kr_encoder_create) against the LayerNorm-kernel path: two handles with the same weights in
ONE process, interleaved rounds (guide rule 24); prints times and the largest difference between the two embeddings.
Usage: python tools/ab_fused.py  (AB_SHAPES=1000x32,1024x128 ...)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kirag_amd import bench_support as BS
dev = torch.device("cuda:0")
encs = {}
for v in ("0", "1"):
    os.environ["KIRAG_AMD_FUSED_LN"] = v
    encs[v] = BS.make_hip_encoder(dev)
shapes = [(1, 32), (2, 256), (8, 128), (125, 32), (1000, 32), (1024, 128), (128, 512)]
if os.environ.get("AB_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["AB_SHAPES"].split(",")]
for B, S in shapes:
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    res = {v: [] for v in encs}
    outs = {}
    for rnd in range(6):
        for v, enc in encs.items():
            for _ in range(2):
                o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 5 if B * S > 20000 else 20
            for _ in range(reps):
                o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / reps * 1e3)
            outs[v] = o
    diff = (outs["0"] - outs["1"]).abs().max().item()
    print(f"{B} x {S}: " + "  ".join(f"fused={v}: median {np.median(res[v]):.3f} ms (min {min(res[v]):.3f})" for v in encs) +
          f"  ratio {np.median(res['1']) / np.median(res['0']):.4f}  max |diff| {diff:.2e}  finite {bool(torch.isfinite(outs['1']).all())}", flush=True)
