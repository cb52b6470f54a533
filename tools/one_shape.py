#!/usr/bin/env python3
"""Runs REPS forwards of one batch shape (python tools/one_shape.py B S [reps]) — a fixed workload to put under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kirag_amd import bench_support as BS
B, S = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
for _ in range(reps):
    enc.forward(ids, mask, 0)
torch.cuda.synchronize()
