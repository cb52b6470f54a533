#!/usr/bin/env python3
"""A/B of search-kernel variants selected by environment variables, interleaved rounds in ONE process on one device (guide rule 24).
Usage: python tools/ab_search.py VAR=a,b [rows] [queries]   e.g. KIRAG_AMD_EPIV=0,1;  VAR1+VAR2=a,b sets both variables to the same value"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kirag_amd.bench_support import CorpusDist
from kirag_amd.retriever.index import FlatIPIndex
var, vals = sys.argv[1].split("=")
vals = vals.split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
d, k = 1024, 100
dev = torch.device("cuda:0")
cd = CorpusDist("gaussian", d, dev)
g = torch.Generator(device=dev); g.manual_seed(3)
ixs = {}
for v in vals:                      # the library reads its switches when an index is created: one index per variant (same rows)
    for one in var.split("+"): os.environ[one] = v
    ixs[v] = FlatIPIndex(d, device=0); ixs[v].reserve(n)
head = None
for s0 in range(0, n, 250_000):
    x = cd.rows(min(250_000, n - s0), g)
    for v in vals: ixs[v].add(x)
    head = x[:nq].clone() if head is None else head
q = cd.queries_near(head, torch.Generator(device=dev).manual_seed(2))
sc = torch.empty((nq, k), dtype=torch.float32, device=dev); rows = torch.empty((nq, k), dtype=torch.int64, device=dev)
res = {v: [] for v in vals}
ref = None
for rnd in range(8):
    for v in vals:
        ix = ixs[v]
        for _ in range(3):
            ix.search_into(q, k, sc, rows)
        cs, ts = [], []
        rr0 = ix.stats()["reranked_rows"]
        for _ in range(5):
            ix.search_into(q, k, sc, rows)
            st = ix.stats(); cs.append(st["last_coarse_ms"]); ts.append(st["last_total_ms"])
        res[v].append((np.median(cs), np.median(ts), (st["reranked_rows"] - rr0) // 5))
        r = rows.cpu().numpy()
        ref = r if ref is None else ref
        assert os.environ.get("AB_NOCHECK") or np.array_equal(r, ref), "variants disagree"   # AB_NOCHECK=1: diagnostic variants with wrong results
for v in vals:
    a = np.array(res[v])
    print(f"{var}={v}: coarse median {np.median(a[:, 0]):.3f} ms (min {a[:, 0].min():.3f}), total median {np.median(a[:, 1]):.3f} ms, re-ranked rows per call {int(a[-1, 2])}; per round {np.round(a[:, 0], 3).tolist()}")
