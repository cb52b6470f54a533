#!/usr/bin/env python3
"""Small-batch forward latency: eager launches vs hipGraph replay (two encoders in one process, interleaved rounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kirag_amd import bench_support as BS
dev = torch.device("cuda:0")
eager = BS.make_hip_encoder(dev)
os.environ["KIRAG_AMD_GRAPH"] = "1"; graph = BS.make_hip_encoder(dev); del os.environ["KIRAG_AMD_GRAPH"]
for B, S in ((1, 32), (2, 256), (4, 64), (8, 128), (32, 32), (100, 32)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    res = {"eager": [], "graph": []}
    for rnd in range(6):
        for name, enc in (("eager", eager), ("graph", graph)):
            for _ in range(3): o = enc.forward(ids, mask, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                o = enc.forward(ids, mask, 0)
                if os.environ.get("LATENCY"): torch.cuda.synchronize()
            torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 30 * 1e3)
    same = torch.equal(eager.forward(ids, mask, 0), graph.forward(ids, mask, 0))
    print(f"{B} x {S}: eager {np.median(res['eager']):.3f} ms, graph {np.median(res['graph']):.3f} ms ({(1 - np.median(res['graph']) / np.median(res['eager'])) * 100:+.1f} % time), identical {same}", flush=True)
