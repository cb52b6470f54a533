#!/usr/bin/env python3
"""Where does Indexer.search_knn spend a 4096-query x top-100 call?  Times the call and its pieces (upload, waits in finish_one, result copies, id gather, string
pass, list building) next to the C-ABI search of the same queries; optional `--one-cpu` pins the process to one CPU first (a slow host).
Usage: python tools/knn_probe.py [rows] [--one-cpu]"""
import os, sys, time
if "--one-cpu" in sys.argv:
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})      # before torch / the library are imported: nothing has touched the GPU yet
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from kirag_amd.retriever import index as I
from kirag_amd.retriever import flat_index as F

rows = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5_000_000
nq, k, d = 4096, 100, 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
ix = I.Indexer(d)
ix.index.reserve(rows)
head = None
for s0 in range(0, rows, 250_000):
    x = torch.nn.functional.normalize(torch.randn(min(250_000, rows - s0), d, generator=g, device=dev), dim=1)
    if head is None:
        head = x[:nq].clone()
    ix.index.add(x); del x
ix.index_id_to_db_id = np.arange(rows, dtype=np.int64) * 3 + 10_000_000_000
q = torch.nn.functional.normalize(head + 0.05 * torch.randn(nq, d, device=dev, generator=g), dim=1).contiguous()
qh = q.cpu().numpy()
ps = torch.empty((nq, k), dtype=torch.float32, pin_memory=True); pi = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)

acc = {}
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def timed(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, timed)
wrap(ix.index, "finish_one", "wait in finish_one")
wrap(ix.index, "search_async", "search_async (enqueue)")
orig_ids = I.ids_to_str_rows
def ids_timed(ext):
    t0 = time.perf_counter(); r = orig_ids(ext); acc["id strings"] = acc.get("id strings", 0.0) + time.perf_counter() - t0; return r
I.ids_to_str_rows = ids_timed
print(f"cpus usable: {len(os.sched_getaffinity(0))}; _fastids extension: {F._fastids is not None}", flush=True)
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.index.search_into(q, k, ps, pi)
    t_abi = time.perf_counter() - t0
    acc.clear(); t0 = time.perf_counter()
    res = ix.search_knn(qh, k, verbose=False)
    t_knn = time.perf_counter() - t0
    if rep:
        other = t_knn - sum(acc.values())
        print(f"search_knn {t_knn * 1e3:.1f} ms vs C ABI {t_abi * 1e3:.1f} ms (ratio {t_abi / t_knn:.3f}): " + ", ".join(f"{k_} {v * 1e3:.1f}" for k_, v in acc.items()) + f", everything else {other * 1e3:.1f} ms", flush=True)
