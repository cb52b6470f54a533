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
stamps = []
_fo = ix.index.finish_one
def _fo_stamped(*a, **kw):
    r = _fo(*a, **kw); stamps.append(time.perf_counter()); return r
ix.index.finish_one = _fo_stamped
wrap(ix.index, "finish_one", "wait in finish_one")
wrap(ix.index, "search_async", "search_async (enqueue)")
orig_ids = I.ids_to_str_rows
id_stamps = []
def ids_timed(ext):
    t0 = time.perf_counter(); r = orig_ids(ext); t1 = time.perf_counter(); acc["id strings"] = acc.get("id strings", 0.0) + t1 - t0; id_stamps.append((t0, t1, len(ext))); return r
I.ids_to_str_rows = ids_timed
print(f"cpus usable: {len(os.sched_getaffinity(0))}; _fastids extension: {F._fastids is not None}", flush=True)
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.index.search_into(q, k, ps, pi)
    t_abi = time.perf_counter() - t0
    res = None                                                      # the previous call's 410 k strings are freed OUTSIDE the timed region (3 ms of deallocation otherwise land in it)
    acc.clear(); stamps.clear(); id_stamps.clear(); t0 = time.perf_counter()
    res = ix.search_knn(qh, k, verbose=False)
    t_knn = time.perf_counter() - t0
    if rep == 4:
        rel = [(s_ - t0) * 1e3 for s_ in stamps]
        print("  id-string passes (start, end, rows): " + " ".join("%.1f-%.1f/%d" % ((a - t0) * 1e3, (b - t0) * 1e3, n_) for a, b, n_ in id_stamps) + f"; return at {t_knn * 1e3:.1f}", flush=True)
        print("  block results final at (ms): " + " ".join("%.1f" % v for v in rel) + "; gaps " + " ".join("%.2f" % (b - a) for a, b in zip([0.0] + rel[:-1], rel)) + f"; tail after the last {t_knn * 1e3 - rel[-1]:.2f} ms", flush=True)
    if rep:
        other = t_knn - sum(acc.values())
        print(f"search_knn {t_knn * 1e3:.1f} ms vs C ABI {t_abi * 1e3:.1f} ms (ratio {t_abi / t_knn:.3f}): " + ", ".join(f"{k_} {v * 1e3:.1f}" for k_, v in acc.items()) + f", everything else {other * 1e3:.1f} ms", flush=True)
