#!/usr/bin/env python3
"""Time of the per-query exact scan (mode = 1) and of a small score_topk (python tools/exact_scan_time.py): A/B with KIRAG_AMD_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kirag_amd.retriever.index import FlatIPIndex
dev = torch.device("cuda:0"); d = 1024
x = torch.nn.functional.normalize(torch.randn(1_000_000, d, device=dev), dim=1)
ix = FlatIPIndex(d, device=0); ix.add(x)
q = torch.nn.functional.normalize(x[:8] + 0.01 * torch.randn(8, d, device=dev), dim=1)
for _ in range(2): ix.search(q, 100, mode=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ix.search(q, 100, mode=1)
torch.cuda.synchronize(); print(f"mode 1, 8 queries x 1M rows: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
small = FlatIPIndex(d, device=0); small.add(x[:1000])
for _ in range(5): small.search(q[:2], 20, mode=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): small.search(q[:2], 20, mode=1)
torch.cuda.synchronize(); print(f"mode 1, 2 queries x 1000 rows: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
