#!/usr/bin/env python3
"""Certificate behaviour of the index on the e5like corpus vs the Gaussian one (GPU box): certified %, re-ranked rows, search time,
and the kernel-independent membership check.  Usage: python tools/e5like_probe.py [rows] [queries]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from kirag_amd.bench_support import CorpusDist
from kirag_amd.retriever.index import FlatIPIndex
import indep_check as IC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d, k = 1024, 100
dev = torch.device("cuda:0")
for kind in ("gaussian", "e5like"):
    cd = CorpusDist(kind, d, dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    ix = FlatIPIndex(d, device=0); ix.reserve(n)
    chunks = []
    for s0 in range(0, n, 250_000):
        m = min(250_000, n - s0)
        x = cd.rows(m, g); ix.add(x)
        if n <= 1_000_000:
            chunks.append((s0, x))
        if s0 == 0:
            head = x[:nq].clone()
    gq = torch.Generator(device=dev); gq.manual_seed(2)
    q = cd.queries_near(head, gq)
    ix.search(q, k); ix.stats(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s, i = ix.search(q, k)
    dt = time.perf_counter() - t0
    st = ix.stats()
    print(f"[{kind}] n={n} nq={nq}: {dt * 1e3:.2f} ms, certified {st['certified']}/{st['queries']}, fallback {st['fallback']}, overflow {st['overflow']}, "
          f"reranked/query {st['reranked_rows'] / max(1, st['queries']):.0f}, coarse {st['last_coarse_ms']:.2f} ms, total {st['last_total_ms']:.2f} ms; "
          f"top1 {s[:, 0].mean():.4f} top100 {s[:, 99].mean():.4f}", flush=True)
    if chunks:
        rs, ri = IC.torch_topk_fp32(q, chunks, k + 32)
        out = IC.check_membership(s, i, rs.cpu().numpy(), ri.cpu().numpy(), k)
        print(f"[{kind}] independent fp32 sgemm + topk check: {out}", flush=True)
    del ix, chunks
    torch.cuda.empty_cache()
