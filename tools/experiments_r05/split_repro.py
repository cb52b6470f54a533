import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from kirag_amd import _lib
from kirag_amd.retriever.index import FlatIPIndex
torch.manual_seed(0)
def run(n, d, nq, k, cuts):
    x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
    pick = torch.randint(0, n, (nq,), device="cuda")
    q = torch.nn.functional.normalize(x[pick] + 0.3 * torch.randn(nq, d, device="cuda") / d ** 0.5, dim=1)
    ix = FlatIPIndex(d); ix.add(x); s0, i0 = ix.search(q, k)
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        sh = FlatIPIndex(d); sh.add(x[a:b]); shards.append((sh, a, b))
    tks = [torch.empty((nq, k + 1), dtype=torch.float32, device="cuda") for _ in shards]
    for (sh, a, b), tk in zip(shards, tks): sh.search_coarse_async(q, k, tk)
    gathered = torch.cat(tks, 0).contiguous()
    sc_all, id_all = [], []
    for (sh, a, b) in shards:
        th = torch.empty((nq,), dtype=torch.float32, device="cuda"); sc = torch.empty((nq, k), dtype=torch.float32, device="cuda"); rw = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        sh.search_global_theta(gathered, len(shards), th); sh.search_rerank_async(th, sc, rw); fl = sh.finish()
        r_ = rw.cpu().numpy(); sc_all.append(sc.cpu().numpy()); id_all.append(np.where(r_ >= 0, r_ + a, -1))
        loc_s, loc_i = sh.search(q, min(k, b - a))
        # every returned row must be a valid row with the right score; every row of the global top-k in this shard must be present
        want = [set(int(v) for v in i0[qi] if a <= v < b) for qi in range(nq)]
        got = [set(int(v) + a for v in r_[qi] if v >= 0) for qi in range(nq)]
        miss = sum(len(w - g) for w, g in zip(want, got))
        print(f"shard [{a},{b}): flagged {fl}, theta[:3] {th[:3].tolist()}, rows returned per query (first 5) {[(r_[qi] >= 0).sum() for qi in range(5)]}, missing global-top-k rows {miss}, tk eps {tks[0][0, k].item():.3e}")
    ms = np.empty((nq, k), np.float32); mi = np.empty((nq, k), np.int64)
    sc_st, id_st = np.ascontiguousarray(np.stack(sc_all)), np.ascontiguousarray(np.stack(id_all))      # named: a temporary would be freed before the call reads it
    _lib.check(_lib.load().kr_topk_merge(sc_st.ctypes.data, id_st.ctypes.data, len(shards), nq, k, ms.ctypes.data, mi.ctypes.data))
    print(n, d, nq, k, cuts, "ids equal", np.array_equal(mi, i0), "scores equal", np.array_equal(ms.view(np.uint32), s0.view(np.uint32)), "bad queries", int((mi != i0).any(1).sum()))
    if not np.array_equal(mi, i0):
        qi = int(np.nonzero((mi != i0).any(1))[0][0]); print(" q", qi, "merged", mi[qi][:8], ms[qi][:4], "want", i0[qi][:8], s0[qi][:4])
run(1003, 512, 300, 100, [0, 115, 1003])
run(1003, 512, 300, 100, [0, 500, 1003])
run(60000, 256, 200, 100, [0, 26000, 60000])
run(5000, 512, 300, 100, [0, 300, 5000])
