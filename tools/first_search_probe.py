import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from kirag_amd.retriever.index import Indexer
n, d = 600_000, 512
g = torch.Generator(device="cuda"); g.manual_seed(31)
x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
src = Indexer(d); src.index_data([str(3 * i) for i in range(n)], x)
folder = tempfile.mkdtemp(); src.serialize(folder)
q = torch.nn.functional.normalize(x[:1] + 0.05 * torch.randn(1, d, device="cuda", generator=g), dim=1).cpu().numpy()
del src, x
ix = Indexer(d); ix.deserialize_from(folder)
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter(); res = ix.search_knn(q, 10, verbose=False); t = (time.perf_counter() - t0) * 1e3
    st = ix.index.stats()
    print(f"search {i}: wall {t:.3f} ms, device total {st['last_total_ms']:.3f} ms, coarse {st['last_coarse_ms']:.3f} ms", flush=True)
# the same after an idle second
time.sleep(1.0)
for i in range(3):
    t0 = time.perf_counter(); res = ix.search_knn(q, 10, verbose=False); t = (time.perf_counter() - t0) * 1e3
    st = ix.index.stats()
    print(f"after 1 s idle, search {i}: wall {t:.3f} ms, device total {st['last_total_ms']:.3f} ms", flush=True)
