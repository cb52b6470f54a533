"""200 blocking searches of one query, top-10, over a resident 5M x 1024 Gaussian corpus (byte pre-scan on): the workload for
rocprofv3 --kernel-trace --stats (per-kernel time of one KiRAG-hop search)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from kirag_amd.bench_support import CorpusDist  # noqa: E402
from kirag_amd.retriever.index import FlatIPIndex  # noqa: E402

N, D = int(os.environ.get("ROWS", 5_000_000)), 1024
dev = torch.device("cuda:0")
cd = CorpusDist(os.environ.get("KIND", "gaussian"), D, dev)
g = torch.Generator(device=dev); g.manual_seed(3)
ix = FlatIPIndex(D, device=0); ix.reserve(N)
head = None
for s0 in range(0, N, 250_000):
    x = cd.rows(min(250_000, N - s0), g); ix.add(x)
    if head is None:
        head = x[:8].clone()
    del x
NQ = int(os.environ.get("NQ", 1))
head = head if NQ <= 8 else torch.cat([head] * 4)
q = cd.queries_near(head[:NQ], g)
for _ in range(200):
    ix.search(q, 10)
print(ix.stats())
