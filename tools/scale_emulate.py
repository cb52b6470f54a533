#!/usr/bin/env python3
"""Per-rank work of `bench.py --gpus W` measured on ONE GPU (no collectives): for W in 1, 2, 4, 8 the rank's share of the strong-scaling job
(5M x 1024 corpus / W rows resident, all 1000 query vectors searched top-100 over the local shard, device merge of W lists + D2H of the result) under the
two encode schedules of bench.py:
  queries  every rank encodes 1000 / W queries of every batch, step by step (round 2)
  batch    blocks of up to W steps: every rank encodes its 1 / W slice of each batch of the block in ONE forward (default; round 4: a partial last block
           of c < W steps is a forward of c * 1000 / W queries per rank instead of a full batch on c ranks and nothing on the others)
The driver runs `--steps 20`: at W = 8 that is two full blocks and one block of 4 steps; the K = 20 line uses the measured forward of 4 * 125 = 500 queries.
The all-gathers cannot be measured on a one-GPU box; they are ESTIMATED (marked est.) as a ring all-gather at 100 GB/s per direction of the
7 x 153 GB/s xGMI links plus 20 us of latency per collective: per step  queries: 4 MB of query vectors + 1.2 MB x W of results;  batch: 4 MB (its share
of the block gather) + 1.2 MB x W; round 5 adds the split search's gather of 0.4 MB x W of coarse scores.  Usage: python tools/scale_emulate.py [total_rows] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kirag_amd import bench_support as BS
from kirag_amd import _lib
from kirag_amd.retriever.index import FlatIPIndex

total = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nq, k, d = 1000, 100, 1024
dev = torch.device("cuda:0")
enc = BS.make_hip_encoder(dev)
ids, mask = BS.synthetic_tokens(dev, nq, 32, seed=2)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


base = None
for world in (1, 2, 4, 8):
    n = (total + world - 1) // world
    g = torch.Generator(device=dev); g.manual_seed(3)
    ix = FlatIPIndex(d, device=0); ix.reserve(n)
    for s0 in range(0, n, 250_000):
        m = min(250_000, n - s0)
        ix.add(torch.nn.functional.normalize(torch.randn(m, d, generator=g, device=dev), dim=1))
    mine = nq // world
    ms_e, _ = timed(lambda: enc.forward(ids[:mine], mask[:mine], 0))
    qv = enc.forward(ids, mask, 0)
    sc = torch.empty((nq, k), dtype=torch.float32, device=dev); rows = torch.empty((nq, k), dtype=torch.int64, device=dev)
    ms_own, _ = timed(lambda: ix.search_into(qv, k, sc, rows))      # every shard certifies and re-ranks its OWN top-k (rounds 2-4)
    coarse = ix.stats()["last_coarse_ms"]
    ms_s = ms_own
    rer_own = rer_split = None
    if world > 1:
        # round 5: the exchange BEFORE the re-rank (kr_index_search_coarse_async -> all-gather of the k best coarse scores -> global bound -> re-rank above it).
        # The W shards of the synthetic corpus are statistically alike, so W copies of this shard's list stand in for the gathered lists: the k-th best of the
        # union of W equal lists is this shard's (k / W)-th best — the bound a real rank would get.  The all-gather itself (nq x (k + 1) floats per rank) is estimated below.
        tk = torch.empty((nq, k + 1), dtype=torch.float32, device=dev); theta = torch.empty((nq,), dtype=torch.float32, device=dev)
        gathered = torch.empty((world * nq, k + 1), dtype=torch.float32, device=dev)

        def split():
            ix.search_coarse_async(qv, k, tk)
            for w in range(world):
                gathered[w * nq:(w + 1) * nq].copy_(tk)
            ix.search_global_theta(gathered, world, theta)
            ix.search_rerank_async(theta, sc, rows)
            ix.finish()
        ix.stats(reset=True); ix.search_into(qv, k, sc, rows); rer_own = ix.stats()["reranked_rows"] / nq
        ix.stats(reset=True); split(); rer_split = ix.stats()["reranked_rows"] / nq
        ms_s, _ = timed(split)
        ix.search_into(qv, k, sc, rows)                                # full local lists for the merge timing below
    # the W gathered lists (here: W copies of the local one, with distinct id ranges) merged on the device + the D2H of the final [nq, k]
    block = (nq * k * 12 + 15) // 16 * 16
    allb = torch.empty(world * block, dtype=torch.uint8, device=dev)
    for w in range(world):
        allb[w * block:w * block + nq * k * 8].view(torch.int64).copy_((rows + w * n).view(-1))
        allb[w * block + nq * k * 8:w * block + nq * k * 12].view(torch.float32).copy_(sc.view(-1))
    out_s = torch.empty_like(sc); out_i = torch.empty_like(rows)
    pin_s = torch.empty((nq, k), dtype=torch.float32, pin_memory=True); pin_i = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)
    lib = _lib.load()

    def merge():
        _lib.check(lib.kr_topk_merge_device(allb.data_ptr() + nq * k * 8, block // 4, allb.data_ptr(), block // 8, world, nq, k,
                                            out_s.data_ptr(), out_i.data_ptr(), 0, torch.cuda.current_stream().cuda_stream))
        pin_s.copy_(out_s, non_blocking=True); pin_i.copy_(out_i, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return pin_s.numpy().copy(), pin_i.numpy().copy()
    ms_m = timed(merge)[0] if world > 1 else 0.0
    if world == 1:
        ms_full = ms_e                                         # the full-batch encode: what a rank pays once per W steps under the batch schedule
    gather = lambda mb: 0.0 if world == 1 else (0.02 + mb * (world - 1) / world / 100.0)      # ms: ring all-gather of `mb` MB in total at 100 GB/s + 20 us
    tk_mb = nq * (k + 1) * 4 / 1e6 * world                      # the extra all-gather of the split search: every rank's k best coarse scores + bound per query
    coll_q = gather(4.0) + gather(1.2 * world) + gather(tk_mb)                  # queries schedule, per step
    coll_b = gather(4.0 * world) / world + gather(1.2 * world) + gather(tk_mb)  # batch schedule, per step
    tot_q = ms_e + ms_s + ms_m
    tot_b = ms_full / world + ms_s + ms_m
    # K steps = K // W full blocks + one block of c = K % W steps, whose encode is ONE forward of c * nq / W queries per rank
    c = K % world
    ms_part = timed(lambda: enc.forward(ids[:c * mine], mask[:c * mine], 0))[0] if c else 0.0
    tot_k = ((K // world) * ms_full + ms_part) / K + ms_s + ms_m
    base_k = globals().setdefault("base_k", tot_k)
    base = base or tot_q
    split_note = "" if world == 1 else f" [own top-k first: {ms_own:.2f} ms, {rer_own:.0f} re-ranked rows per query; exchange first: {rer_split:.0f}]"
    print(f"W={world}: encode {mine} queries {ms_e:.2f} ms | search 1000 x {n} rows {ms_s:.2f} ms (coarse {coarse:.2f}){split_note} | device merge + D2H of the result {ms_m:.2f} ms\n"
          f"      queries schedule: {tot_q:.2f} ms per step (x{base / tot_q:.2f}); with est. collectives {tot_q + coll_q:.2f} ms (x{base / (tot_q + coll_q):.2f})\n"
          f"      batch schedule:   full-batch encode {ms_full:.2f} ms per {world} steps -> {tot_b:.2f} ms per step (x{base / tot_b:.2f}); "
          f"with est. collectives {tot_b + coll_b:.2f} ms (x{base / (tot_b + coll_b):.2f})\n"
          f"      batch schedule, K = {K} steps ({K // world} full blocks + one of {c}: forward of {c * mine} queries {ms_part:.2f} ms): {tot_k:.2f} ms per step "
          f"(x{base_k / tot_k:.2f}); with est. collectives {tot_k + coll_b:.2f} ms (x{base_k / (tot_k + coll_b):.2f})", flush=True)
    del ix
    torch.cuda.empty_cache()
