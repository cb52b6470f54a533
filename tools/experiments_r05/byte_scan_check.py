"""Byte pre-scan of small query blocks: the same exact top-k with kr_set_option("byte_prescan", 0 / 1), rows marked, time per search.
usage: python3 tools/experiments_r05/byte_scan_check.py [rows] [kinds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from kirag_amd import _lib  # noqa: E402
from kirag_amd.bench_support import CorpusDist  # noqa: E402
from kirag_amd.retriever.index import FlatIPIndex  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["gaussian", "e5like"]
D = 1024
lib = _lib.load()
dev = torch.device("cuda:0")


def opt(v):
    _lib.check(lib.kr_set_option(b"byte_prescan", v))


def timed(ix, q, k, reps=30):
    ix.search(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.search(q, k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for kind in KINDS:
    cd = CorpusDist(kind, D, dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    ix = FlatIPIndex(D, device=0); ix.reserve(N)
    head = None
    for s0 in range(0, N, 250_000):
        x = cd.rows(min(250_000, N - s0), g); ix.add(x)
        if head is None:
            head = x[:64].clone()
        del x
    gq = torch.Generator(device=dev); gq.manual_seed(2)
    for nq in (1, 2, 8, 16, 32):
        qn = cd.queries_near(head[:nq], gq)
        qr = cd.rows(nq, gq)                       # queries that aim at nothing in particular
        for name, q in (("near", qn), ("free", qr)):
            for k in (10, 100):
                opt(0); ix.stats(reset=True)
                s0_, i0_ = ix.search(q, k); st0 = ix.stats(reset=True)
                opt(1)
                s1_, i1_ = ix.search(q, k); st1 = ix.stats(reset=True)
                same = np.array_equal(i0_, i1_) and np.array_equal(np.asarray(s0_).view(np.uint32), np.asarray(s1_).view(np.uint32))
                opt(0); t0 = timed(ix, q, k)
                opt(1); t1 = timed(ix, q, k)
                print(f"[{kind} {N}] nq {nq} {name} k {k:3d}: same {same} | 16-bit {t0:.3f} ms (cert {st0['certified']}/{nq}) | byte {t1:.3f} ms (cert {st1['certified']}/{nq}, "
                      f"scans {st1['byte_scans']}, marked {st1['byte_marked_rows']}, coarse {st1['last_coarse_ms']:.3f} ms)", flush=True)
                assert same
    del ix
    torch.cuda.empty_cache()
