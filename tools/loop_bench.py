#!/usr/bin/env python3
"""BASELINE config 5 micro-benchmark: one KiRAG retrieval turn on ONE GPU with synthetic inputs =
    encode nq in {1,2} chain queries (256 tokens) + T=512 triples (32 tokens; cached after the first turn) -> exact top-20 of [nq]x[T] on the device
    + exact top-10 search of the nq query vectors over the resident 5M x 1024 corpus.
Prints per-stage latencies (ms).  Usage: python tools/loop_bench.py [total_rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kirag_amd import bench_support as BS
from kirag_amd.retriever.index import FlatIPIndex
from kirag_amd.retriever.aligner import rank_by_similarity

total = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
ix = FlatIPIndex(1024, device=0); ix.reserve(total)
for s0 in range(0, total, 250_000):
    m = min(250_000, total - s0)
    ix.add(torch.nn.functional.normalize(torch.randn(m, 1024, generator=g, device=dev), dim=1))
enc = BS.make_hip_encoder(dev)
q_ids, q_mask = BS.synthetic_tokens(dev, 2, 256, seed=2)
t_ids, t_mask = BS.synthetic_tokens(dev, 512, 32, seed=4)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

for nq in (1, 2):
    ms_q, qv = timed(lambda: enc.forward(q_ids[:nq], q_mask[:nq], 0))
    ms_t, tv = timed(lambda: enc.forward(t_ids, t_mask, 0))
    ms_r, _ = timed(lambda: rank_by_similarity(qv, tv, 20))
    ms_s, _ = timed(lambda: ix.search(qv, 10))
    st = ix.stats(reset=True)
    byte = st.get("byte_scans", 0) > 0          # the final coarse round streamed the int8 copy (1 KiB per row) instead of the 16-bit one (2 KiB)
    streamed = total * (1024 if byte else 2048)
    print(f"nq={nq}: encode queries {ms_q:.2f} ms | encode 512 triples {ms_t:.2f} ms (first turn only with the cache) | rank top-20 {ms_r:.2f} ms | "
          f"search top-10 over {total} rows {ms_s:.2f} ms (coarse {st['last_coarse_ms']:.2f} ms = {streamed / st['last_coarse_ms'] / 1e9:.2f} TB/s of the "
          f"{'int8' if byte else '16-bit'} copy's bytes{', %d rows marked per search' % (st['byte_marked_rows'] // max(1, st['byte_scans'])) if byte else ''})")
