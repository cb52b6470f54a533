"""Where Indexer.search_knn's time goes at 4096 queries x top-100 over 5M rows (cProfile of the surface call next to the C-ABI time)."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from kirag_amd.bench_support import CorpusDist
from kirag_amd.retriever.index import FlatIPIndex, Indexer
N, D, nq, k = 5_000_000, 1024, 4096, 100
dev = torch.device("cuda:0")
cd = CorpusDist("gaussian", D, dev)
g = torch.Generator(device=dev); g.manual_seed(3)
index = FlatIPIndex(D, device=0); index.reserve(N)
head = None
for s0 in range(0, N, 250_000):
    x = cd.rows(min(250_000, N - s0), g); index.add(x)
    if head is None:
        head = x[:nq].clone()
    del x
q = cd.queries_near(head, g).contiguous(); q_host = q.cpu().numpy()
ix = Indexer.__new__(Indexer); ix.faiss_padding = False; ix.index = index
ix.index_id_to_db_id = np.arange(N, dtype=np.int64) * 3 + 10_000_000_000
ps = torch.empty((nq, k), dtype=torch.float32, pin_memory=True); pi = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)
for _ in range(2):
    index.search_into(q, k, ps, pi); ix.search_knn(q_host, k, verbose=False)
ta, tb = [], []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); index.search_into(q, k, ps, pi); ta.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ix.search_knn(q_host, k, verbose=False); tb.append(time.perf_counter() - t0)
print("C ABI %.2f ms, search_knn %.2f ms (medians of 5), ratio %.3f" % (np.median(ta) * 1e3, np.median(tb) * 1e3, np.median(ta) / np.median(tb)))
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    ix.search_knn(q_host, k, verbose=False)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
