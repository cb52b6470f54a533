#!/usr/bin/env python3
"""Encode-leg micro benchmark: passages/s of the HIP encoder at the e5-large shape (synthetic weights)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kirag_amd import bench_support as BS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
ids, mask = BS.synthetic_tokens(dev, n, S, seed=1, ragged=ragged)
for _ in range(2):
    enc.forward(ids, mask, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    enc.forward(ids, mask, 0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = BS.encoder_flops(enc.cfg, mask.sum(1))
print(f"tile={os.environ.get('KIRAG_AMD_PROJ_TILE','auto')} n={n} S={S} ragged={ragged}: {dt*1e3:.2f} ms  {n/dt:.0f} seq/s  {fl/dt/1e12:.0f} TFLOP/s")
