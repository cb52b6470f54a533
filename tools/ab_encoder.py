#!/usr/bin/env python3
"""A/B of encoder variants selected by an environment variable, interleaved rounds in ONE process (guide rule 24).
Usage: python tools/ab_encoder.py VAR=a,b   (batch shapes: 125x32, 250x32, 1000x32, 1024x128, 64x512)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kirag_amd import bench_support as BS
var, vals = sys.argv[1].split("=")
vals = vals.split(",")
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
shapes = [(1, 32), (2, 256), (4, 64), (8, 128), (32, 32), (100, 32)] if os.environ.get("AB_TINY") else [(125, 32), (150, 32), (250, 32), (64, 64), (32, 128), (1000, 32)] if os.environ.get("AB_SMALL") else [(125, 32), (250, 32), (1000, 32), (8, 128), (1024, 128), (128, 512)]
if os.environ.get("AB_SHAPES"):       # e.g. AB_SHAPES=1000x32,1024x128
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["AB_SHAPES"].split(",")]
for B, S in shapes:
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    res = {v: [] for v in vals}
    outs = {}
    for rnd in range(6):
        for v in vals:
            os.environ[var] = v
            for _ in range(2):
                o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 5 if B * S > 20000 else 20
            for _ in range(reps):
                o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / reps * 1e3)
            outs[v] = o
    same = all(torch.equal(outs[vals[0]], outs[v]) for v in vals)
    print(f"{B} x {S}: " + "  ".join(f"{var}={v}: median {np.median(res[v]):.3f} ms (min {min(res[v]):.3f})" for v in vals) + f"  outputs identical: {same}", flush=True)
