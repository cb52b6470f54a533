"""GPU parity tests of the index path (through the C ABI, via kirag_amd.retriever.index) against the oracle.
Bar: internal rows identical and scores bit-identical to oracle.search_np.search_canonical (the exact inner product rounded once to fp32);
at BASELINE sizes membership is checked against a kernel-independent fp32 sgemm + top-k (tests/indep_check.py)."""
import os
import sys

import numpy as np
import pytest

from oracle import search_np as S

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import indep_check as IC  # noqa: E402

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _mk(d, x, **kw):
    from kirag_amd.retriever.index import Indexer
    ix = Indexer(d, **kw)
    ix.index_data([str(i) for i in range(len(x))], x)
    return ix


def _queries_near(rng, x, nq, noise=0.05):
    pick = rng.choice(len(x), nq, replace=len(x) < nq)
    q = x[pick] + noise * rng.standard_normal((nq, x.shape[1])).astype(np.float32) / np.sqrt(x.shape[1]) * 8
    return (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32), pick


@pytest.mark.parametrize("n,d,nq,k", [(1000, 64, 32, 10), (1000, 1024, 32, 10), (300, 128, 5, 300), (4097, 256, 17, 100),
                                      (129, 1024, 3, 1), (5000, 768, 40, 20)])
def test_small_exact_vs_canonical(n, d, nq, k):
    rng = np.random.default_rng(n + d)
    x = _unit(rng, n, d)
    q, _ = _queries_near(rng, x, nq)
    ix = _mk(d, x)
    s, i = ix.index.search(q, k)
    so, io = S.search_canonical(q, x, k)
    assert np.array_equal(i, io)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32))
    # exact-scan mode must agree too
    s2, i2 = ix.index.search(q, k, mode=1)
    assert np.array_equal(i2, io) and np.array_equal(s2.view(np.uint32), so.view(np.uint32))


@pytest.mark.parametrize("n,d,nq,k", [(40000, 1024, 2, 10), (30000, 768, 9, 50), (25000, 512, 32, 100), (20000, 384, 1, 10), (33333, 1024, 31, 1)])
def test_few_queries_stream_scan_vs_canonical(n, d, nq, k):
    """At most 32 queries with d in {384, 512, 768, 1024}: the register-resident-queries stream kernel (k_coarse_q32; 32-row tile slots, several
    rounds, wave-private LDS-DMA rings), bit-exact vs the C oracle, certified, and identical to the exact-scan mode."""
    rng = np.random.default_rng(n + d + nq)
    x = _unit(rng, n, d)
    x[n - 5] = x[3]                                             # a duplicate row far away: tie broken by row index
    q, pick = _queries_near(rng, x, nq)
    ix = _mk(d, x)
    s, i = ix.index.search(q, k)
    so, io = S.search_canonical(q, x, k)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    st = ix.index.stats()
    assert st["queries"] == nq and st["certified"] + st["fallback"] == nq and st["certified"] >= nq - 1, st
    s2, i2 = ix.index.search(q, k, mode=1)
    assert np.array_equal(i2, io) and np.array_equal(s2.view(np.uint32), so.view(np.uint32))


def test_multi_round_vs_canonical_config1_shape():
    """20k x 1024, 200 queries, top-100: two coarse rounds + certified re-rank, bit-exact vs the C oracle."""
    rng = np.random.default_rng(7)
    x = _unit(rng, 20000, 1024)
    q, pick = _queries_near(rng, x, 200)
    ix = _mk(1024, x)
    s, i = ix.index.search(q, 100)
    so, io = S.search_canonical(q, x, 100)
    assert np.array_equal(i, io)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32))
    assert np.array_equal(i[:, 0], pick)
    st = ix.index.stats()
    assert st["queries"] == 200 and st["certified"] + st["fallback"] == 200
    assert st["certified"] >= 190, st      # the certificate must hold for (nearly) all random-data queries


@pytest.mark.parametrize("coarse", ["bf16", "f16"])
def test_three_rounds_vs_f64_and_exact_mode(coarse):
    rng = np.random.default_rng(8)
    x = _unit(rng, 150_000, 256)
    q, pick = _queries_near(rng, x, 64)
    ix = _mk(256, x, coarse_dtype=coarse)
    s, i = ix.index.search(q, 100)
    so, io = S.search_f64(q, x, 100)
    assert np.array_equal(i, io)
    np.testing.assert_array_equal(s, so)
    s1, i1 = ix.index.search(q[:8], 100, mode=1)
    assert np.array_equal(i1, io[:8]) and np.array_equal(s1, so[:8])


def test_search_knn_surface_ids_and_block_boundary():
    rng = np.random.default_rng(9)
    x = _unit(rng, 3000, 128)
    ids = [str(10_000_000_000 + 3 * j) for j in range(3000)]      # int64-only ids, returned as str
    from kirag_amd.retriever.index import Indexer
    ix = Indexer(128)
    ix.index_data(ids[:1000], x[:1000]); ix.index_data(ids[1000:], x[1000:].astype(np.float64))
    assert ix.index.ntotal == 3000 and ix.index_id_to_db_id.dtype == np.int64
    q = _unit(rng, 1030, 128)                                     # crosses the 1024-query block (index.py:39-46)
    res = ix.search_knn(q, 7, verbose=False)
    so, io = S.search_f64(q, x, 7)
    assert len(res) == 1030
    for r in (0, 511, 1023, 1024, 1029):
        assert res[r][0] == [ids[j] for j in io[r]] and all(isinstance(v, str) for v in res[r][0])
        assert res[r][1].dtype == np.float32 and np.array_equal(res[r][1], so[r])
    with pytest.raises(ValueError):
        ix.search_knn(q[:2], 3001, verbose=False)
    with pytest.raises(NotImplementedError):
        Indexer(128, metric="l2")


def test_search_knn_pipelined_blocks_equal_the_block_by_block_form():
    """Round 5: ``Indexer.search_knn`` enqueues block i + 1 before it converts block i's ids (two pinned result slots, ``kr_index_search_async`` /
    ``finish``) and builds the id strings in bulk (``kr_format_ids``).  Three and a half blocks, a small ``index_batch_size``, queries that pass 1 cannot
    certify (duplicated rows: the finish patches the pinned slot), negative ids, a torch tensor as input — every list identical to the reference's
    loop (index.py:36-53) evaluated block by block through ``index.search`` + ``str``."""
    import torch
    from kirag_amd.retriever.index import Indexer
    rng = np.random.default_rng(19)
    x = _unit(rng, 5000, 128)
    x[4000:4600] = x[17]                                            # 601 identical rows: whoever asks for row 17 gets mass ties (passes 2 / 3)
    ids = [str(v) for v in (np.arange(5000, dtype=np.int64) * 7 - 9000)]   # negative and positive int64 ids
    ix = Indexer(128)
    ix.index_data(ids, x)
    q = _unit(rng, 3 * 1024 + 500, 128)
    q[5] = x[17]; q[2047] = x[17]; q[3500] = x[17]
    for bs, k in ((1024, 10), (300, 100), (1024, 1)):
        res = ix.search_knn(q, k, index_batch_size=bs, verbose=False)
        assert len(res) == len(q)
        for s0 in range(0, len(q), bs):
            so, io = ix.index.search(q[s0:s0 + bs], k)
            ext = ix.index_id_to_db_id[io]
            for r in range(len(io)):
                assert res[s0 + r][0] == [str(v) for v in ext[r].tolist()], (bs, k, s0 + r)
                assert res[s0 + r][1].dtype == np.float32 and np.array_equal(res[s0 + r][1], so[r]), (bs, k, s0 + r)
    res_t = ix.search_knn(torch.from_numpy(q), 10, verbose=False)   # tensor input takes the same path
    res_n = ix.search_knn(q, 10, verbose=False)
    assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(res_t, res_n))
    assert ix.index.stats()["fine"] + ix.index.stats()["exact"] > 0     # the uncertified queries really went through passes 2 / 3


@pytest.mark.parametrize("nq,bs", [(2600, 1024), (1024 + 512, 1024), (1300, 600), (3000, 1000)])
def test_search_knn_last_block_in_two_pieces_and_two_calls_in_flight(nq, bs):
    """Indexer.search_knn with full-size blocks: the last block of >= 512 queries is searched as 3/4 + 1/4 (whole 128-query tiles), two blocks are in
    flight at any time; every row of every query must equal the one-shot search (ids mapped through index_id_to_db_id, scores bit for bit)."""
    rng = np.random.default_rng(nq + bs)
    n, d, k = 20000, 128, 7
    x = _unit(rng, n, d)
    q, _ = _queries_near(rng, x, nq)
    from kirag_amd.retriever.index import Indexer
    ix = Indexer(d)
    ids = [str(10_000_000_000 + 3 * i) for i in range(n)]
    ix.index_data(ids, x)
    res = ix.search_knn(q, k, index_batch_size=bs, verbose=False)
    s, i = ix.index.search(q, k)
    assert len(res) == nq
    for r in range(nq):
        assert res[r][0] == [ids[j] for j in i[r]], r
        assert np.array_equal(np.asarray(res[r][1]).view(np.uint32), s[r].view(np.uint32)), r
    assert ix.index._lib.kr_index_search_pending(ix.index._h) == 0


def test_finish_one_finishes_the_oldest_call_only():
    """ABI 8: kr_index_search_finish_one — two asynchronous searches in flight, the oldest is finished (its results are final) while the newer one stays
    outstanding; a third call on a handle with nothing outstanding is a no-op."""
    import torch
    from kirag_amd.retriever.index import FlatIPIndex
    rng = np.random.default_rng(5)
    x = _unit(rng, 30000, 256)
    q, _ = _queries_near(rng, x, 40)
    ix = FlatIPIndex(256, device=0); ix.add(x)
    so, io = S.search_canonical(q, x, 10)
    qa, qb = torch.from_numpy(q[:20]).cuda(), torch.from_numpy(q[20:]).cuda()
    outs = [(torch.empty((20, 10), dtype=torch.float32, pin_memory=True), torch.empty((20, 10), dtype=torch.int64, pin_memory=True)) for _ in range(2)]
    ix.search_async(qa, 10, *outs[0]); ix.search_async(qb, 10, *outs[1])
    assert ix._lib.kr_index_search_pending(ix._h) == 2
    assert ix.finish_one() == 0 and ix._lib.kr_index_search_pending(ix._h) == 1
    assert np.array_equal(outs[0][1].numpy(), io[:20]) and np.array_equal(outs[0][0].numpy().view(np.uint32), so[:20].view(np.uint32))
    assert ix.finish_one() == 0 and ix._lib.kr_index_search_pending(ix._h) == 0
    assert np.array_equal(outs[1][1].numpy(), io[20:]) and np.array_equal(outs[1][0].numpy().view(np.uint32), so[20:].view(np.uint32))
    assert ix.finish_one() == 0


def test_split_search_exchange_before_rerank_two_shards_in_one_process():
    """Round 5: kr_index_search_coarse_async -> [gather of the shards' k best coarse scores] -> kr_index_search_global_theta ->
    kr_index_search_rerank_async.  Two row shards of one corpus held by two indexes of ONE process, the gather done by hand: the merged lists must be the
    unsharded canonical answer bit for bit, every shard must re-rank FEWER rows than its stand-alone search (that is the point) and return padded tails;
    theta = None must reproduce kr_index_search_async exactly; a first half without its second half is reported by finish."""
    import torch
    from kirag_amd import _lib
    from kirag_amd.retriever.index import FlatIPIndex
    rng = np.random.default_rng(23)
    n, d, nq, k = 60000, 256, 200, 100
    x = _unit(rng, n, d)
    x[n // 2 + 3] = x[5]                                            # a tie across the shard boundary
    q = (x[rng.choice(n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    so, io = S.search_canonical(q, x, k)
    cuts = [0, 26000, n]                                            # uneven shards
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        ix = FlatIPIndex(d); ix.add(torch.from_numpy(x[a:b]).cuda()); shards.append((ix, a))
    qd = torch.from_numpy(q).cuda()
    # stand-alone searches: the rows each shard re-ranks when it certifies its own top-k
    alone = []
    for ix, a in shards:
        ix.stats(reset=True); ix.search(qd, k); alone.append(ix.stats()["reranked_rows"])
    # split form
    tks = [torch.empty((nq, k + 1), dtype=torch.float32, device="cuda") for _ in shards]
    for (ix, a), tk in zip(shards, tks):
        ix.stats(reset=True)
        ix.search_coarse_async(qd, k, tk)
    gathered = torch.cat(tks[::-1], dim=0).contiguous()            # "any rank order"
    outs = []
    for (ix, a) in shards:
        theta = torch.empty((nq,), dtype=torch.float32, device="cuda")
        sc = torch.empty((nq, k), dtype=torch.float32, device="cuda"); rw = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        ix.search_global_theta(gathered, len(shards), theta)
        ix.search_rerank_async(theta, sc, rw)
        assert ix.finish() == [0]                                   # one call outstanding, every query certified
        outs.append((sc.cpu().numpy(), rw.cpu().numpy(), a))
    split_rows = [ix.stats()["reranked_rows"] for ix, _ in shards]
    assert all(s_ < 0.6 * a_ for s_, a_ in zip(split_rows, alone)), (split_rows, alone)
    sc_all = np.stack([o[0] for o in outs]); id_all = np.stack([np.where(o[1] >= 0, o[1] + o[2], -1) for o in outs])
    assert (id_all < 0).any() and np.isneginf(sc_all[id_all < 0]).all()      # shards really returned fewer than k rows for some queries
    for o in outs:                                                  # padding only at the tail, lists sorted
        valid = o[1] >= 0
        assert (valid[:, :-1] >= valid[:, 1:]).all()
    ms = np.empty((nq, k), np.float32); mi = np.empty((nq, k), np.int64)
    sc_all, id_all = np.ascontiguousarray(sc_all), np.ascontiguousarray(id_all)                  # named: the buffers must outlive the call
    _lib.check(_lib.load().kr_topk_merge(sc_all.ctypes.data, id_all.ctypes.data, len(shards), nq, k, ms.ctypes.data, mi.ctypes.data))
    assert np.array_equal(mi, io) and np.array_equal(ms.view(np.uint32), so.view(np.uint32))
    # theta = None: the shard's own answer, exactly kr_index_search_async's
    ix, a = shards[0]
    tk = tks[0]
    sc = torch.empty((nq, k), dtype=torch.float32, device="cuda"); rw = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ix.search_coarse_async(qd, k, tk); ix.search_rerank_async(None, sc, rw); ix.finish()
    s1, i1 = ix.search(qd, k)
    assert np.array_equal(rw.cpu().numpy(), i1) and np.array_equal(sc.cpu().numpy().view(np.uint32), s1.view(np.uint32))
    # a dropped second half
    ix.search_coarse_async(qd, k, tk)
    with pytest.raises(_lib.KiragAmdError, match="rerank"):
        ix.finish()
    s2, i2 = ix.search(qd, k)                                       # the handle is usable again
    assert np.array_equal(i2, i1)


def test_global_theta_of_eight_shard_blocks_with_padding_vs_a_numpy_restatement():
    """kr_index_search_global_theta at the node's world size without the node: the gathered [8][nq][k + 1] blocks of eight shards - this
    shard's own block (from kr_index_search_coarse_async), synthetic blocks for the others, one of them all -inf (a shard that had fewer than k candidates for
    a query, as k_local_topk writes it), one with a larger error bound, in arbitrary rank order - against a numpy restatement of the bound
    theta[q] = (k-th largest of the 8 k coarse scores) - (max error bound + this shard's own) * 1.000001, in fp32; then the eight-shard search itself: eight
    row shards of one corpus in ONE process, split form with the gather done by hand, merged == the unsharded canonical answer bit for bit."""
    import torch
    from kirag_amd import _lib
    from kirag_amd.retriever.index import FlatIPIndex
    rng = np.random.default_rng(88)
    n, d, nq, k, W = 48_000, 256, 33, 20, 8
    x = _unit(rng, n, d)
    q = (x[rng.choice(n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    qd = torch.from_numpy(q).cuda()
    cuts = [0, 300, 9000, 9000 + 25, 20000, 26000, 33000, 41000, n]          # uneven: one shard of 25 rows (>= k), one of 300
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        ix = FlatIPIndex(d); ix.add(torch.from_numpy(x[a:b]).cuda()); shards.append((ix, a))
    # (1) the bound, on shard 4's handle, from synthetic neighbours
    ix4 = shards[4][0]
    tk = torch.empty((nq, k + 1), dtype=torch.float32, device="cuda")
    ix4.search_coarse_async(qd, k, tk)
    own = tk.cpu().numpy()
    blocks = [own]
    for w in range(1, W):
        b = np.sort(rng.uniform(-0.2, 0.9, (nq, k)).astype(np.float32), axis=1)[:, ::-1].copy()
        e = np.full((nq, 1), 1e-3 * w, np.float32)
        if w == 3:
            b[:] = -np.inf                                             # a shard without k candidates for any query
        if w == 5:
            b[::2] = -np.inf; e[:] = 0.05                              # ... for every other query, and the largest error bound of all
        blocks.append(np.concatenate([b, e], axis=1))
    order = [6, 0, 3, 7, 5, 1, 4, 2]
    gathered = torch.from_numpy(np.concatenate([blocks[w] for w in order], axis=0)).cuda().contiguous()
    theta = torch.empty((nq,), dtype=torch.float32, device="cuda")
    ix4.search_global_theta(gathered, W, theta)
    allsc = np.concatenate([blk[:, :k] for blk in blocks], axis=1)         # [nq, W * k]
    bg = np.sort(allsc, axis=1)[:, ::-1][:, k - 1].astype(np.float32)
    emax = np.max(np.stack([blk[:, k] for blk in blocks]), axis=0).astype(np.float32)
    want = (bg - ((emax + own[:, k]).astype(np.float32) * np.float32(1.000001)).astype(np.float32)).astype(np.float32)
    assert np.array_equal(theta.cpu().numpy().view(np.uint32), want.view(np.uint32))
    sc = torch.empty((nq, k), dtype=torch.float32, device="cuda"); rw = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ix4.search_rerank_async(theta, sc, rw); ix4.finish()                   # the call is completed (a synthetic bound: the rows are not looked at)
    # (2) the eight-shard split search, gather by hand
    tks = []
    for ix, a in shards:
        t = torch.empty((nq, k + 1), dtype=torch.float32, device="cuda"); ix.search_coarse_async(qd, k, t); tks.append(t)
    gathered = torch.cat([tks[w] for w in order], dim=0).contiguous()
    outs = []
    for ix, a in shards:
        th = torch.empty((nq,), dtype=torch.float32, device="cuda")
        sc = torch.empty((nq, k), dtype=torch.float32, device="cuda"); rw = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        ix.search_global_theta(gathered, W, th); ix.search_rerank_async(th, sc, rw); ix.finish()
        r = rw.cpu().numpy()
        outs.append((sc.cpu().numpy(), np.where(r >= 0, r + a, -1)))
    sc_all = np.ascontiguousarray(np.stack([o[0] for o in outs])); id_all = np.ascontiguousarray(np.stack([o[1] for o in outs]))
    ms = np.empty((nq, k), np.float32); mi = np.empty((nq, k), np.int64)
    _lib.check(_lib.load().kr_topk_merge(sc_all.ctypes.data, id_all.ctypes.data, W, nq, k, ms.ctypes.data, mi.ctypes.data))
    so, io = S.search_canonical(q, x, k)
    assert np.array_equal(mi, io) and np.array_equal(ms.view(np.uint32), so.view(np.uint32))
    assert (id_all < 0).mean() > 0.5                                       # most of the 8 x k slots are padding: each shard re-ranked only what can be in the global top-k


def test_duplicates_tie_rule_and_k_equals_n():
    rng = np.random.default_rng(10)
    x = _unit(rng, 500, 64)
    x[100] = x[7]; x[333] = x[7]; x[499] = x[7]
    ix = _mk(64, x)
    s, i = ix.index.search(x[[7]], 5)
    assert list(i[0, :4]) == [7, 100, 333, 499] and len(set(s[0, :4].tolist())) == 1
    s_all, i_all = ix.index.search(x[[7, 8]], 500)
    so, io = S.search_canonical(x[[7, 8]], x, 500)
    assert np.array_equal(i_all, io) and np.array_equal(s_all, so)


def test_dense_cluster_forces_exact_fallback():
    """Rows closer together than the bf16 error bound: the certificate must refuse and the exact scan answer."""
    rng = np.random.default_rng(11)
    base = _unit(rng, 1, 256)
    x = base + 2e-4 * rng.standard_normal((6000, 256)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    q = _unit(rng, 6, 256) * 0.2 + base
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    ix = _mk(256, x)
    s, i = ix.index.search(q, 50)
    so, io = S.search_canonical(q, x, 50)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    st = ix.index.stats()
    assert st["fallback"] > 0 and st["fine"] == st["fallback"] and st["exact"] == 0, st      # the fp64 pass certifies what bf16 cannot


def test_nan_rows_are_never_returned():
    rng = np.random.default_rng(12)
    x = _unit(rng, 700, 64)
    x[5] = np.nan                                                  # an all-masked passage encodes to NaN (encoders.py:56-58)
    q = _unit(rng, 4, 64)
    ix = _mk(64, x)
    s, i = ix.index.search(q, 20)
    assert not (i == 5).any() and np.isfinite(s).all()
    keep = np.ones(700, bool); keep[5] = False
    so, io = S.search_canonical(q, x[keep], 20)
    remap = np.nonzero(keep)[0]
    assert np.array_equal(i, remap[io]) and np.array_equal(s, so)


def test_shard_merge_equals_unsharded_on_gpu():
    import ctypes as C
    from kirag_amd import _lib
    rng = np.random.default_rng(13)
    x = _unit(rng, 9000, 128); q = _unit(rng, 33, 128)
    x[8000] = x[50]
    full = _mk(128, x)
    s, i = full.index.search(q, 30)
    bounds = [0, 2000, 4500, 9000]
    ss, ii = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        sh = _mk(128, x[a:b])
        s_, i_ = sh.index.search(q, 30)
        ss.append(s_); ii.append(i_ + a)
    sc = np.ascontiguousarray(np.stack(ss)); ic = np.ascontiguousarray(np.stack(ii))
    ms = np.empty((33, 30), np.float32); mi = np.empty((33, 30), np.int64)
    _lib.check(_lib.load().kr_topk_merge(sc.ctypes.data, ic.ctypes.data, 3, 33, 30, ms.ctypes.data, mi.ctypes.data))
    assert np.array_equal(mi, i) and np.array_equal(ms, s)
    om, oi = S.merge_shards(ss, ii, 30)
    assert np.array_equal(oi, mi) and np.array_equal(om, ms)


def test_device_merge_equals_host_merge():
    """kr_topk_merge_device (rank-by-binary-search over the gathered lists in HBM) == kr_topk_merge, with ties across shards, padded
    (short) shards, an all-padding shard and strided list blocks (the [ids | scores] byte blocks of ShardedSearcher)."""
    import torch
    from kirag_amd import _lib
    from kirag_amd.parallel import merge_topk
    lib = _lib.load()
    rng = np.random.default_rng(21)
    for W, nq, k in ((8, 300, 100), (3, 17, 1), (2, 5, 1024), (5, 40, 37)):
        sc = np.round(rng.standard_normal((W, nq, k)).astype(np.float32), 1)
        ids = np.stack([rng.permuted(np.tile(np.arange(w * 100_000, w * 100_000 + max(2 * k, 64)), (nq, 1)), axis=1)[:, :k] for w in range(W)]).astype(np.int64)
        order = np.lexsort((ids, -sc), axis=2)
        sc = np.take_along_axis(sc, order, 2); ids = np.take_along_axis(ids, order, 2)
        if k > 1:
            sc[1, :, k // 2:] = -np.inf; ids[1, :, k // 2:] = -1          # a short shard
        if W > 2:
            sc[2, 0] = -np.inf; ids[2, 0] = -1                              # nothing from shard 2 for query 0
        hs, hi = merge_topk(sc, ids, k)
        block = (nq * k * 12 + 15) // 16 * 16
        buf = torch.zeros(W * block, dtype=torch.uint8)
        for w in range(W):
            buf[w * block:w * block + nq * k * 8] = torch.from_numpy(ids[w].reshape(-1).view(np.uint8))
            buf[w * block + nq * k * 8:w * block + nq * k * 12] = torch.from_numpy(sc[w].reshape(-1).view(np.uint8))
        dbuf = buf.cuda()
        os_ = torch.empty((nq, k), dtype=torch.float32, device="cuda"); oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        _lib.check(lib.kr_topk_merge_device(dbuf.data_ptr() + nq * k * 8, block // 4, dbuf.data_ptr(), block // 8, W, nq, k,
                                            os_.data_ptr(), oi.data_ptr(), 0, None))
        torch.cuda.synchronize()
        assert np.array_equal(oi.cpu().numpy(), hi) and np.array_equal(os_.cpu().numpy().view(np.uint32), hs.view(np.uint32))
    with pytest.raises(_lib.KiragAmdError):
        _lib.check(lib.kr_topk_merge_device(dbuf.data_ptr(), 1, dbuf.data_ptr(), 1, 9, 1, 1024, os_.data_ptr(), oi.data_ptr(), 0, None))


def test_serialize_roundtrip(tmp_path):
    rng = np.random.default_rng(14)
    x = _unit(rng, 1234, 64)
    from kirag_amd.retriever.index import Indexer
    ix = Indexer(64); ix.index_data([str(5 * j) for j in range(1234)], x)
    ix.serialize(str(tmp_path))
    iy = Indexer(64); iy.deserialize_from(str(tmp_path))
    assert iy.index.ntotal == 1234 and np.array_equal(iy.index_id_to_db_id, ix.index_id_to_db_id)
    assert np.array_equal(iy.index.reconstruct_n(0, 1234), x)
    q = _unit(rng, 3, 64)
    a = ix.search_knn(q, 5, verbose=False); b = iy.search_knn(q, 5, verbose=False)
    assert all(a[r][0] == b[r][0] and np.array_equal(a[r][1], b[r][1]) for r in range(3))


def test_k_200_uses_large_candidate_buffers():
    """k > 102 switches to K1 = 512 / cap = 8192 (64 KiB + of dynamic LDS in k_select / k_rerank)."""
    rng = np.random.default_rng(21)
    x = _unit(rng, 30000, 128); q, _ = _queries_near(rng, x, 9)
    ix = _mk(128, x)
    s, i = ix.index.search(q, 200)
    so, io = S.search_f64(q, x, 200)
    assert np.array_equal(i, io) and np.array_equal(s, so)
    s, i = ix.index.search(q, 600)            # K1 = 1024, cap = 8192, doubling rounds
    so, io = S.search_canonical(q, x, 600)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    s, i = ix.index.search(q, 1024)           # the largest k: K1 = 2048
    so, io = S.search_canonical(q, x, 1024)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    st = ix.index.stats()
    assert st["exact"] == 0, st               # round 1 sent every k > 204 to the per-query exact scan


def test_seeded_random_shape_sweep_vs_canonical():
    """30 seeded random (n, d, nq, k) combinations — d any multiple of 4 (dpad padding), nq across the 256-query tile and the 1024-query block
    boundary, k up to the fast path's limit and beyond (exact-scan path), clustered data (dense near-ties) — all bit-exact vs the oracle."""
    rng = np.random.default_rng(2026)
    for case in range(30):
        d = int(rng.choice([4, 20, 64, 100, 256, 384, 768, 1024, 1500]))
        n = int(rng.integers(1, 9000))
        nq = int(rng.choice([1, 2, 3, 31, 255, 256, 257, 700, 1025, 1100])) if case % 3 == 0 else int(rng.integers(1, 300))
        k = int(min(n, rng.choice([1, 5, 10, 20, 100, 204, 205, 400])))
        x = _unit(rng, n, d)
        if case % 4 == 1 and n > 50:                       # a tight cluster: many scores within the 16-bit resolution
            x[: n // 2] = x[0] + 1e-3 * rng.standard_normal((n // 2, d)).astype(np.float32)
            x /= np.linalg.norm(x, axis=1, keepdims=True)
        q, _ = _queries_near(rng, x, nq)
        ix = _mk(d, x)
        s, i = ix.index.search(q, k)
        so, io = S.search_canonical(q, x, k)
        assert np.array_equal(i, io), (case, n, d, nq, k)
        assert np.array_equal(s.view(np.uint32), so.view(np.uint32)), (case, n, d, nq, k)


def test_many_rows_small_dim_tile_permutation_64bit():
    """24M rows x d=4: 93750 corpus tiles, so (tile slot x permutation multiplier) exceeds 2^32 and the multi-round schedule runs 5 rounds:
    exercises the tile-coordinate arithmetic of the coarse scan at a row count no d=1024 shard reaches.  Each query has 50 planted rows
    (scaled copies of the query, scattered over the whole row range incl. the last tile) far above the background (norm 0.5), so the fast path
    must certify: a tile that the permutation skipped or visited twice would lose / duplicate a planted row.  Bit-exact vs the oracle."""
    rng = np.random.default_rng(77)
    n, d, nq, k = 24_000_000, 4, 8, 50
    x = rng.standard_normal((n, d), dtype=np.float32)
    x *= (0.5 / np.linalg.norm(x, axis=1, keepdims=True))
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    pos = rng.choice(n - 1, nq * k - 1, replace=False).reshape(-1)
    pos = np.concatenate([pos, [n - 1]]).reshape(nq, k)            # one planted row is the very last row
    for qi in range(nq):
        for j in range(k):
            x[pos[qi, j]] = q[qi] * np.float32(1.0 - 1e-3 * j)
    from kirag_amd.retriever.index import FlatIPIndex
    ix = FlatIPIndex(d)
    ix.reserve(n)
    for s0 in range(0, n, 4_000_000):
        ix.add(x[s0:s0 + 4_000_000])
    s, i = ix.search(q, k)
    so, io = S.search_canonical(q, x, k)
    st = ix.stats()
    print(f"[24M x 4] certified {st['certified']} fallback {st['fallback']} rounds {st['coarse_rounds']}")
    assert np.array_equal(i, io)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32))
    assert np.array_equal(np.sort(i, axis=1), np.sort(pos, axis=1))
    assert st["certified"] == nq and st["fallback"] == 0 and st["coarse_rounds"] == 5


# ---------------------------------------------------------------------------------------------------------------------------------------
# canonical score = exact inner product rounded once: known answers from Python rationals, and the rare exact path forced
# ---------------------------------------------------------------------------------------------------------------------------------------
def _score_topk1(q, x):
    """one (q, x) pair through kr_score_topk (k_exact_scan): the canonical score as the library computes it"""
    import ctypes as C
    from kirag_amd import _lib
    d = len(q); dp = max(4, (d + 3) // 4 * 4)
    qq = np.zeros((1, dp), np.float32); xx = np.zeros((1, dp), np.float32)
    qq[0, :d] = q; xx[0, :d] = x
    sc = np.empty((1, 1), np.float32); rows = np.empty((1, 1), np.int64)
    _lib.check(_lib.load().kr_score_topk(qq.ctypes.data, 1, xx.ctypes.data, 1, dp, 1, sc.ctypes.data, rows.ctypes.data, 0, None))
    return sc[0, 0]


@pytest.mark.parametrize("force", [0, 1])
def test_canonical_score_golden_vectors_on_gpu(golden, force):
    """tests/golden/g9_exact_dot.npz (351 known answers from Python rationals: exact midpoints, sticky bits, subnormal and vanishing results,
    overflow to inf, cancellation, 120-bit exponent spreads) through the library's scoring kernel, with the certified fp64 fast path (force = 0)
    and with every score sent through the integer super-accumulator (force = 1): every bit must match."""
    from kirag_amd import _lib
    lib = _lib.load()
    g = golden("g9_exact_dot.npz")
    off = g["offsets"]
    _lib.check(lib.kr_set_option(b"force_exact_scores", force))
    try:
        for c in range(len(off) - 1):
            q = g["q_bits"][off[c]:off[c + 1]].view(np.float32); x = g["x_bits"][off[c]:off[c + 1]].view(np.float32)
            got = np.float32(_score_topk1(q, x)).view(np.uint32)
            assert got == g["expected_bits"][c], (c, len(q), hex(int(got)), hex(int(g["expected_bits"][c])))
    finally:
        _lib.check(lib.kr_set_option(b"force_exact_scores", 0))
    with pytest.raises(_lib.KiragAmdError):
        _lib.check(lib.kr_set_option(b"no_such_option", 1))


def test_forced_exact_path_equals_fast_path_in_search():
    """Every re-rank / exact-scan score through the super-accumulator: same rows, same score bits as the fast path and as the oracle."""
    from kirag_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(31)
    x = _unit(rng, 6000, 1024); q, _ = _queries_near(rng, x, 20)
    x2 = (_unit(rng, 3000, 100) * np.float32(37.5)).astype(np.float32); q2 = (_unit(rng, 7, 100) * np.float32(1e-3)).astype(np.float32)
    for xx, qq, d, k in ((x, q, 1024, 100), (x2, q2, 100, 10)):
        ix = _mk(d, xx)
        s0, i0 = ix.index.search(qq, k)
        so, io = S.search_canonical(qq, xx, k)
        _lib.check(lib.kr_set_option(b"force_exact_scores", 1))
        try:
            s1, i1 = ix.index.search(qq, k)
            s2, i2 = ix.index.search(qq[:3], k, mode=1)
        finally:
            _lib.check(lib.kr_set_option(b"force_exact_scores", 0))
        for s_, i_ in ((s0, i0), (s1, i1)):
            assert np.array_equal(i_, io) and np.array_equal(s_.view(np.uint32), so.view(np.uint32))
        assert np.array_equal(i2, io[:3]) and np.array_equal(s2.view(np.uint32), so[:3].view(np.uint32))


# ---------------------------------------------------------------------------------------------------------------------------------------
# pass 2 (fp64 MFMA scan of the fp32 rows for queries the 16-bit scan cannot certify) and pass 3 (exact scan)
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,nq,k", [(1000, 64, 5, 10), (4097, 256, 17, 100), (20000, 1024, 33, 100), (70000, 1024, 16, 20), (9000, 100, 40, 204),
                                      (50000, 2048, 9, 10), (33, 4, 3, 33), (30000, 768, 70, 300)])
def test_fine_pass_only_vs_canonical(n, d, nq, k):
    """mode = 2: no 16-bit scan, every query through the fp64 MFMA pass (groups of 32 / 16 queries, one and two 16-query tiles, several rounds,
    a partial last 32-row slot, d not a multiple of 128), bit-exact vs the C oracle; nothing may need the per-query exact scan."""
    rng = np.random.default_rng(n + d + nq)
    x = _unit(rng, n, d)
    if n > 100:
        x[n - 3] = x[5]                                                # a duplicate row: tie broken by row number
    q, _ = _queries_near(rng, x, nq)
    ix = _mk(d, x)
    s, i = ix.index.search(q, k, mode=2)
    so, io = S.search_canonical(q, x, k)
    assert np.array_equal(i, io), (n, d, nq, k)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32))
    st = ix.index.stats()
    assert st["fine"] == nq and st["exact"] == 0 and st["certified"] == 0, st


def test_near_duplicate_corpus_all_queries_through_the_fine_pass():
    """A corpus of near-duplicates (pairwise score differences ~1e-5, far below the bf16 bound but above the fp64 pass's): every query fails the
    16-bit certificate; 300 flagged queries = 10 groups of 32 sharing one pass over the rows each; bit-exact vs the oracle."""
    rng = np.random.default_rng(41)
    base = _unit(rng, 1, 512)
    x = base + 3e-5 * rng.standard_normal((40000, 512)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    q = (_unit(rng, 300, 512) * 0.3 + base)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    ix = _mk(512, x)
    s, i = ix.index.search(q, 100)
    so, io = S.search_canonical(q, x, 100)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    st = ix.index.stats()
    print(f"[near-duplicates] {st}")
    assert st["fallback"] >= 290 and st["fine"] == st["fallback"] and st["exact"] == 0, st


def test_mass_exact_ties_reach_the_exact_scan():
    """5000 identical rows: more exact ties at the k-th score than the certified re-rank can hold, for both passes -> pass 3 (exact scan);
    ties come back in row order."""
    rng = np.random.default_rng(42)
    x = _unit(rng, 8000, 128)
    x[1000:6000] = x[999]
    q = x[[999, 7000]]
    ix = _mk(128, x)
    s, i = ix.index.search(q, 50)
    so, io = S.search_canonical(q, x, 50)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    assert list(i[0]) == list(range(999, 1049))
    st = ix.index.stats()
    assert st["exact"] >= 1 and st["fine"] + st["exact"] == st["fallback"], st


def test_fine_pass_top_1024_of_4096():
    """k = 1024 (K1 = 2048, 8192-entry buffers, doubling rounds) through the fp64 MFMA pass: a quarter of the corpus returned in exact order."""
    rng = np.random.default_rng(43)
    x = _unit(rng, 4096, 1024); q = _unit(rng, 32, 1024)
    ix = _mk(1024, x)
    s, i = ix.index.search(q, 1024, mode=2)
    so, io = S.search_canonical(q, x, 1024)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))


# ---------------------------------------------------------------------------------------------------------------------------------------
# BASELINE sizes: kernel-independent membership check (fp32 sgemm + torch.topk), Gaussian and e5like corpora
# ---------------------------------------------------------------------------------------------------------------------------------------
def _build_resident(n, d, kind, keep_chunks, chunk=250_000):
    import torch
    from kirag_amd.bench_support import CorpusDist
    from kirag_amd.retriever.index import FlatIPIndex
    dev = torch.device("cuda:0")
    cd = CorpusDist(kind, d, dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    ix = FlatIPIndex(d, device=0); ix.reserve(n)
    chunks, head = [], None
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        x = cd.rows(m, g); ix.add(x)
        if head is None:
            head = x[:1000].clone()
        if keep_chunks:
            chunks.append((s0, x))
        del x
    return ix, cd, chunks, head


def _regen_chunks(n, d, kind, chunk=250_000):
    """the same corpus again, chunk by chunk (rows are a pure function of the generator state): for sizes where a second resident copy is too much"""
    import torch
    from kirag_amd.bench_support import CorpusDist
    dev = torch.device("cuda:0")
    cd = CorpusDist(kind, d, dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    for s0 in range(0, n, chunk):
        yield s0, cd.rows(min(chunk, n - s0), g)


@pytest.mark.parametrize("kind", ["gaussian", "e5like"])
def test_config2_1M_all_queries_vs_independent_topk_and_oracle(kind):
    """BASELINE config 2 (1M x 1024, 1000 queries, top-100) on the Gaussian corpus and on the e5like one (scores in a narrow band around 0.75,
    rank-100 gaps ~5e-5): ALL 1000 queries against torch's fp32 sgemm + topk (no missed row, no wrong row, scores within 2e-6), 32 queries bit-exact
    against the C oracle over the full corpus, planted neighbour first, certified fraction reported."""
    import torch
    n, d, nq, k = 1_000_000, 1024, 1000, 100
    ix, cd, chunks, head = _build_resident(n, d, kind, keep_chunks=True)
    gq = torch.Generator(device="cuda"); gq.manual_seed(2)
    q = cd.queries_near(head[:nq], gq)
    s, i = ix.search(q, k)
    st = ix.stats()
    print(f"[1M {kind}] certified {st['certified']}/{nq}, fine {st['fine']}, exact {st['exact']}, reranked/query {st['reranked_rows'] / nq:.0f}")
    assert np.array_equal(i[:, 0], np.arange(nq))                                 # planted neighbour first
    assert (np.diff(s, axis=1) <= 0).all()
    rs, ri = IC.torch_topk_fp32(q, chunks, k + 32)
    out = IC.check_membership(s, i, rs.cpu().numpy(), ri.cpu().numpy(), k)
    assert out["queries"] == nq
    assert st["certified"] >= 0.95 * nq and st["exact"] == 0, st
    xh = torch.cat([c[1] for c in chunks]).cpu().numpy()
    so, io = S.search_canonical(q[:32].cpu().numpy(), xh, k)
    assert np.array_equal(i[:32], io) and np.array_equal(s[:32].view(np.uint32), so.view(np.uint32))
    # shard-merge == unsharded (two 500k shards), through the host merge
    from kirag_amd import _lib
    from kirag_amd.retriever.index import FlatIPIndex
    halves = []
    for a, b in ((0, 500_000), (500_000, n)):
        sh = FlatIPIndex(d); sh.reserve(b - a)
        for r0, xc in chunks:
            if a <= r0 < b:
                sh.add(xc)
        s_, i_ = sh.search(q[:64], k); halves.append((s_, i_ + a)); del sh
    sc2 = np.ascontiguousarray(np.stack([h[0] for h in halves])); ic2 = np.ascontiguousarray(np.stack([h[1] for h in halves]))
    ms = np.empty((64, k), np.float32); mi = np.empty((64, k), np.int64)
    _lib.check(_lib.load().kr_topk_merge(sc2.ctypes.data, ic2.ctypes.data, 2, 64, k, ms.ctypes.data, mi.ctypes.data))
    assert np.array_equal(mi, i[:64]) and np.array_equal(ms, s[:64])


@pytest.mark.parametrize("kind", ["gaussian", "e5like"])
def test_metric_size_5M_all_queries_vs_independent_topk(kind):
    """The metric's own size (5M x 1024 on one GPU, 1000 queries, top-100; BASELINE config 3's corpus): all 1000 queries against the
    kernel-independent fp32 sgemm + topk (the corpus is regenerated chunk by chunk for the reference), 8 queries bit-exact against the C oracle
    on the rows of the reference's top-132 lists plus 200k further rows."""
    import torch
    n, d, nq, k = 5_000_000, 1024, 1000, 100
    ix, cd, _, head = _build_resident(n, d, kind, keep_chunks=False)
    gq = torch.Generator(device="cuda"); gq.manual_seed(2)
    q = cd.queries_near(head[:nq], gq)
    s, i = ix.search(q, k)
    st = ix.stats()
    print(f"[5M {kind}] certified {st['certified']}/{nq}, fine {st['fine']}, exact {st['exact']}, reranked/query {st['reranked_rows'] / nq:.0f}, "
          f"coarse {st['last_coarse_ms']:.2f} ms, total {st['last_total_ms']:.2f} ms")
    assert np.array_equal(i[:, 0], np.arange(nq)) and (np.diff(s, axis=1) <= 0).all()
    rs, ri = IC.torch_topk_fp32(q, _regen_chunks(n, d, kind), k + 32)
    out = IC.check_membership(s, i, rs.cpu().numpy(), ri.cpu().numpy(), k)
    assert out["queries"] == nq and st["certified"] >= 0.95 * nq and st["exact"] == 0, (out, st)
    # canonical score bits of the returned rows (C oracle on the gathered rows)
    rows = i[:8].reshape(-1)
    xs = ix.reconstruct_rows(rows) if hasattr(ix, "reconstruct_rows") else np.stack([ix.reconstruct_n(int(r), 1)[0] for r in rows])
    sc = S.scores_at(q[:8].cpu().numpy(), xs, np.arange(8 * k).reshape(8, k).astype(np.int64))
    assert np.array_equal(sc.view(np.uint32), s[:8].view(np.uint32))
    # the block schedule of bench.py --gpus N at the metric's size: three batches enqueued back to back (ABI 5: several asynchronous searches outstanding on
    # one stream, nothing waits for the device), finished together — each bit-identical to the blocking search of its batch
    qs = [torch.roll(q, r, dims=0).contiguous() for r in range(3)]
    outs = [(torch.empty((nq, k), dtype=torch.float32, device="cuda"), torch.empty((nq, k), dtype=torch.int64, device="cuda")) for _ in qs]
    for qq, (sa, ia) in zip(qs, outs):
        ix.search_async(qq, k, sa, ia)
    assert ix._lib.kr_index_search_pending(ix._h) == 3
    assert len(ix.finish()) == 3 and ix._lib.kr_index_search_pending(ix._h) == 0
    for r, (sa, ia) in enumerate(outs):
        assert np.array_equal(ia.cpu().numpy(), np.roll(i, r, axis=0)) and np.array_equal(sa.cpu().numpy().view(np.uint32), np.roll(s, r, axis=0).view(np.uint32))


@pytest.mark.parametrize("kind", ["gaussian", "e5like"])
def test_metric_size_5M_small_blocks_take_the_byte_prescan_and_meet_the_independent_topk(kind):
    """The int8 pre-scan at the metric's size against a formulation that shares nothing with it (only nq <= 2 met the independent
    top-k at 5M rows): blocks of 1 / 8 / 32 queries, top-10 and top-100, over 5M x 1024 rows — every search must have taken the pre-scan (byte_scans), every
    query is compared with the fp32 sgemm + topk over the regenerated corpus, the score bits of the 8-query block with the C oracle on the gathered rows, and
    every result equals the same search with the pre-scan switched off bit for bit.  (e5like blocks of 32 queries mark more than n / 8 rows: still exact, and
    after four such blocks the index pauses the path - the feedback the state-machine test below drives on purpose.)"""
    import torch
    n, d = 5_000_000, 1024
    ix, cd, _, head = _build_resident(n, d, kind, keep_chunks=False)
    gq = torch.Generator(device="cuda"); gq.manual_seed(2)
    q_all = cd.queries_near(head[:41], gq)
    cases = [(q_all[0:1], 10), (q_all[1:9], 100), (q_all[9:41], 10)]
    got = []
    _opt(b"byte_prescan", 1); _opt(b"debug_byte_min_rows", -1)
    try:
        for q, k in cases:
            q = q.contiguous()
            before = ix.stats()["byte_scans"]
            s, i = ix.search(q, k)
            st = ix.stats()
            assert st["byte_scans"] == before + 1 and st["byte_rows"] == n and st["exact"] == 0, (len(q), k, st)
            _opt(b"byte_prescan", 0)
            s0, i0 = ix.search(q, k)
            _opt(b"byte_prescan", 1)
            assert ix.stats()["byte_scans"] == before + 1
            assert np.array_equal(i, i0) and np.array_equal(s.view(np.uint32), s0.view(np.uint32)), (len(q), k)
            got.append((s, i))
            print(f"[5M {kind}] nq={len(q)} k={k}: pre-scan marked {st['byte_marked_rows']} rows so far, total {st['last_total_ms']:.2f} ms")
    finally:
        _opt(b"byte_prescan", 1)
    rs, ri = IC.torch_topk_fp32(q_all.contiguous(), _regen_chunks(n, d, kind), 100 + 32)
    rs, ri = rs.cpu().numpy(), ri.cpu().numpy()
    lo = 0
    for (q, k), (s, i) in zip(cases, got):
        out = IC.check_membership(s, i, rs[lo:lo + len(q), :k + 32], ri[lo:lo + len(q), :k + 32], k)
        assert out["queries"] == len(q)
        lo += len(q)
    s, i = got[1]
    xs = ix.reconstruct_rows(i.reshape(-1))
    sc = S.scores_at(q_all[1:9].cpu().numpy(), xs, np.arange(i.size).reshape(i.shape).astype(np.int64))
    assert np.array_equal(sc.view(np.uint32), s.view(np.uint32))


def test_config4_size_21M_search_only_vs_independent_topk():
    """BASELINE config 4's corpus size on ONE GPU: 21,015,324 x 1024 (43 GB bf16 + 86 GB fp32 resident), 256 queries, top-100, search only;
    every query against the kernel-independent fp32 sgemm + topk over the regenerated corpus."""
    import torch
    n, d, nq, k = 21_015_324, 1024, 256, 100
    ix, cd, _, head = _build_resident(n, d, "gaussian", keep_chunks=False, chunk=500_000)
    gq = torch.Generator(device="cuda"); gq.manual_seed(2)
    q = cd.queries_near(head[:nq], gq)
    s, i = ix.search(q, k)
    st = ix.stats()
    print(f"[21M] certified {st['certified']}/{nq}, fine {st['fine']}, exact {st['exact']}, coarse {st['last_coarse_ms']:.2f} ms, total {st['last_total_ms']:.2f} ms")
    assert np.array_equal(i[:, 0], np.arange(nq)) and (np.diff(s, axis=1) <= 0).all()
    rs, ri = IC.torch_topk_fp32(q, _regen_chunks(n, d, "gaussian", chunk=500_000), k + 32)
    out = IC.check_membership(s, i, rs.cpu().numpy(), ri.cpu().numpy(), k)
    assert out["queries"] == nq and st["exact"] == 0, (out, st)


def test_pass2_prescan_marks_only_the_slots_that_matter():
    """A boilerplate cluster inside an ordinary corpus (5000 near-duplicates of one row among 60k Gaussian rows, scattered over the row range): queries that
    aim at the cluster cannot be certified by the 16-bit pass; pass 2 first marks, with the 16-bit stream kernel and pass 1's own bound b_k - 2 eps, the ROWS
    that can still matter and runs its fp64 scan over that list only.  Results must equal the oracle bit for bit, with and without the pre-scan, and the other
    queries must certify in pass 1."""
    import os
    rng = np.random.default_rng(51)
    n, d = 60_000, 1024
    x = _unit(rng, n, d)
    where = rng.choice(n, 5000, replace=False)                          # more rows inside the 16-bit bound than the certified re-rank holds
    base = _unit(rng, 1, d)
    x[where] = base + 3e-5 * rng.standard_normal((5000, d)).astype(np.float32)
    x[where] /= np.linalg.norm(x[where], axis=1, keepdims=True)
    q_plain, _ = _queries_near(rng, x, 40)
    q_aim = base + 0.2 * _unit(rng, 24, d)
    q_aim = (q_aim / np.linalg.norm(q_aim, axis=1, keepdims=True)).astype(np.float32)
    q = np.concatenate([q_aim, q_plain]).astype(np.float32)
    so, io = S.search_canonical(q, x, 100)
    ix = _mk(d, x)
    s, i = ix.index.search(q, 100)
    st = ix.index.stats()
    print(f"[prescan] {st}")
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    assert st["fine"] >= 24 and st["exact"] == 0 and st["marked_passes"] >= 1 and st["certified"] >= 30, st
    assert 5000 <= st["marked_rows"] / st["marked_passes"] <= 5600, st          # the cluster and little else
    assert set(i[0].tolist()) <= set(where.tolist())                         # a cluster query's top-100 are cluster rows
    os.environ["KIRAG_AMD_NO_MARK"] = "1"                                   # the switches are read once, when an index is created
    try:
        ix2 = _mk(d, x)
    finally:
        del os.environ["KIRAG_AMD_NO_MARK"]
    s2, i2 = ix2.index.search(q, 100)
    assert np.array_equal(i2, io) and np.array_equal(s2.view(np.uint32), so.view(np.uint32))
    st2 = ix2.index.stats()
    assert st2["marked_passes"] == 0 and st2["fine"] == st["fine"], st2



# ---------------------------------------------------------------------------------------------------------------------------------------
# byte pre-scan of small query blocks (ABI 8): the final coarse round through the int8 copy
# ---------------------------------------------------------------------------------------------------------------------------------------
def _opt(name, v):
    from kirag_amd import _lib
    _lib.check(_lib.load().kr_set_option(name, v))


@pytest.fixture()
def byte_everywhere():
    """small blocks take the byte pre-scan at every index size (the product default is 2^19 rows)"""
    _opt(b"debug_byte_min_rows", 0); _opt(b"byte_prescan", 1)
    yield
    _opt(b"debug_byte_min_rows", -1); _opt(b"byte_prescan", 1)


@pytest.mark.parametrize("n,d,nq,k", [(40000, 1024, 2, 10), (30011, 768, 8, 50), (25000, 512, 1, 100), (33333, 1024, 7, 1), (50000, 1000, 3, 20), (9000, 768, 5, 64), (40000, 1024, 32, 10), (30000, 768, 17, 100), (26000, 512, 9, 5), (45000, 384, 4, 10), (20000, 380, 1, 30)])
def test_byte_prescan_small_corpora_vs_canonical(byte_everywhere, n, d, nq, k):
    """Blocks of <= 32 queries with the int8 final round forced on at small sizes: bit-exact vs the C oracle (rows and score bits), identical to the same
    search with the option off, ties by row index across the byte-scanned region, and the statistics show that the path ran and what it marked."""
    rng = np.random.default_rng(n + d + nq)
    x = _unit(rng, n, d)
    q, pick = _queries_near(rng, x, nq)
    x[n - 5] = x[pick[0]]; x[n // 2 + 7] = x[pick[0]]            # duplicates of the best row far away: the tie is broken by row index, every copy must be marked
    ix = _mk(d, x)
    s, i = ix.index.search(q, k)
    st = ix.index.stats(reset=True)
    so, io = S.search_canonical(q, x, k)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    assert st["byte_scans"] == 1 and 0 < st["byte_marked_rows"] < n, st
    assert st["certified"] + st["fallback"] == nq
    _opt(b"byte_prescan", 0)
    s0, i0 = ix.index.search(q, k)
    st0 = ix.index.stats(reset=True)
    assert st0["byte_scans"] == 0 and np.array_equal(i0, i) and np.array_equal(s0.view(np.uint32), s.view(np.uint32))


def test_byte_prescan_f16_coarse_copy_and_row_shards_in_split_form(byte_everywhere):
    """The f16 flavour of the 16-bit copy behind the byte round (k_score_list<F16>), and the split row-sharded search (coarse half -> gathered coarse scores ->
    global bound -> re-rank half) with the byte round inside each shard's coarse half: merged lists = the canonical answer."""
    import torch
    from kirag_amd import _lib
    from kirag_amd.retriever.index import FlatIPIndex
    rng = np.random.default_rng(4242)
    n, d, nq, k = 36000, 1024, 3, 10
    x = _unit(rng, n, d)
    q, pick = _queries_near(rng, x, nq)
    so, io = S.search_canonical(q, x, k)
    ix = _mk(d, x, coarse_dtype="f16")
    s, i = ix.index.search(q, k)
    st = ix.index.stats(reset=True)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32)) and st["byte_scans"] == 1, st
    cuts = [0, 15000, n]
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        sh = FlatIPIndex(d, device=0); sh.add(x[a:b]); shards.append((sh, a))
    qd = torch.from_numpy(q).cuda()
    tks = [torch.empty((nq, k + 1), dtype=torch.float32, device="cuda") for _ in shards]
    for (sh, a), tk in zip(shards, tks):
        sh.search_coarse_async(qd, k, tk)
    gathered = torch.cat(tks, dim=0).contiguous()
    sc_all, id_all = [], []
    for (sh, a) in shards:
        th = torch.empty((nq,), dtype=torch.float32, device="cuda")
        sc = torch.empty((nq, k), dtype=torch.float32, device="cuda"); rw = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        sh.search_global_theta(gathered, len(shards), th); sh.search_rerank_async(th, sc, rw); sh.finish()
        assert sh.stats()["byte_scans"] == 1
        r_ = rw.cpu().numpy(); sc_all.append(sc.cpu().numpy()); id_all.append(np.where(r_ >= 0, r_ + a, -1))
    ms = np.empty((nq, k), np.float32); mi = np.empty((nq, k), np.int64)
    sc_st, id_st = np.ascontiguousarray(np.stack(sc_all)), np.ascontiguousarray(np.stack(id_all))      # named: they must outlive the call
    _lib.check(_lib.load().kr_topk_merge(sc_st.ctypes.data, id_st.ctypes.data, len(shards), nq, k, ms.ctypes.data, mi.ctypes.data))
    assert np.array_equal(mi, io) and np.array_equal(ms.view(np.uint32), so.view(np.uint32))


def _row_built_to_reach_the_byte_bound():
    """Random data leaves the byte bound ~30 x slack, so a too-small eps8 would pass every other test.  Here a row of the exact top-k is BUILT so that its byte
    score underestimates its exact score by > 99.9 % of |u| |r - r^|, the Cauchy-Schwarz term that dominates eps8 (k_scan8_prep): the query has equal-magnitude
    components (the inequality is tight), every component of the row sits 1/64 of a step short of a rounding boundary on the side that loses against the query's
    sign (31/64 of a step lost per component, against the 1/2 the worst case allows: the built row also sets max |r - r^| over the index), one component pins the
    row's scale to a power of two, the first 65 536 rows come in +- pairs (centre exactly 0, unit weights).  k - 1 anchors and one further anchor with exact
    scores just below the built row's sit in the tiles of the 16-bit rounds, the built row in a tile of the byte round: it must displace the lowest anchor.
    Returns (x, q, k, built row, anchor rows)."""
    from math import gcd
    rng = np.random.default_rng(20260)
    n, d, k = 400_000, 1024, 10
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[1:65536:2] = -x[0:65536:2]                                   # mean of the first 65 536 rows: exactly zero in any summation order of +- pairs
    sign = np.where(rng.random(d) < 0.5, -1.0, 1.0).astype(np.float32)
    q = (sign / np.float32(32.0)).astype(np.float32)[None, :]      # |q_i| = 1 / sqrt(d) exactly, |q| = 1
    # which 32-row tiles the two 16-bit rounds visit (run_rounds: slots [0, 4096) of the multiplicative tile permutation for cap = 4096, K1 = 64)
    ntiles = (n + 31) // 32
    mul = max(1, int(ntiles * 0.6180339887498949))
    while gcd(mul, ntiles) != 1:
        mul += 1
    sample_tiles = {(s * mul) % ntiles for s in range(4096)}
    late_sample = sorted(t for t in sample_tiles if t * 32 >= 65536)
    byte_tiles = [t for t in range(ntiles) if t not in sample_tiles and t * 32 >= 65536]
    S0 = 0.5
    anchor_rows = [late_sample[7 * j + 3] * 32 + 5 for j in range(k)]
    for j, r in enumerate(anchor_rows):                            # exact scores S0, S0 + 1e-3, ..., S0 + 9e-3 (unit rows: score = cosine with q)
        c = S0 + 1e-3 * j
        z = rng.standard_normal(d).astype(np.float64); z -= (z @ q[0].astype(np.float64)) * q[0]; z /= np.linalg.norm(z)
        x[r] = (c * q[0].astype(np.float64) + np.sqrt(1.0 - c * c) * z).astype(np.float32)
    # the built row: r_i = sign(q_i) s (m_i + 31/64): the rounding error loses against q_i everywhere; scale pinned by one component at 127 s
    s_ = np.float64(2.0 ** -10)
    target = S0 + 5e-4                                             # between the lowest and the second-lowest anchor
    m = np.floor(rng.random(d) * 40.0)                             # magnitudes (m_i + 31/64) s with the sign of q_i: the value rounds to sign m s, so the
    frac = 31.0 / 64.0                                             # rounding error sign (31/64) s has the sign of q_i on EVERY component (and is exact in fp32)
    mag = (m + frac) * s_
    row = sign.astype(np.float64) * mag
    row[0] = sign[0] * 127.0 * s_
    # scale the integer parts so that q.row hits the target (keep the fractional parts): adjust m on a few components
    cur = float((q[0].astype(np.float64) * row).sum())
    need = target - cur
    steps = int(round(need * 32.0 / s_))                           # one unit of m on one component changes the score by s / 32
    idx = 1
    while steps != 0:
        stp = 1 if steps > 0 else -1
        if 0 <= m[idx] + stp <= 120:
            m[idx] += stp; steps -= stp
        idx = 1 + (idx % (d - 1))
    mag = (m + frac) * s_
    row = sign.astype(np.float64) * mag
    row[0] = sign[0] * 127.0 * s_
    built = byte_tiles[len(byte_tiles) // 2] * 32 + 11
    x[built] = row.astype(np.float32)
    assert np.array_equal(x[built].astype(np.float64), row)        # every component is exact in fp32
    exact_built = float((q[0].astype(np.float64) * row).sum())
    err = row - s_ * np.rint(row / s_)                             # what the int8 copy loses of the built row
    under = float((q[0].astype(np.float64) * err).sum())
    assert abs(exact_built - target) < 2e-5 and under > 0.999 * float(np.linalg.norm(err)) > 0.013, (exact_built, under, float(np.linalg.norm(err)))
    return x, q, k, built, anchor_rows


def test_byte_prescan_bound_holds_for_a_row_built_to_reach_it(byte_everywhere):
    """The real bound (debug_eps8_permille = 1000, the default) marks the built row: rows and score bits equal the C oracle's."""
    x, q, k, built, anchor_rows = _row_built_to_reach_the_byte_bound()
    ix = _mk(x.shape[1], x)
    s, i = ix.index.search(q, k)
    st = ix.index.stats(reset=True)
    so, io = S.search_canonical(q, x, k)
    assert st["byte_scans"] == 1, st
    assert built in io[0] and anchor_rows[0] not in io[0]          # the oracle agrees with the construction
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32)), (i, io)


def test_byte_prescan_mutant_bound_misses_the_built_row(byte_everywhere):
    """The test above has teeth (that claim used to live in a commit message): the SAME index searched with the pre-scan's bound scaled to
    750 permille (kr_set_option "debug_eps8_permille", a mutation hook) does NOT mark the built row — the lowest anchor stays in the result, which is therefore
    wrong — while 1000 permille, set explicitly before and after, returns the oracle's rows.  The pre-scan ran in all three searches (byte_scans)."""
    x, q, k, built, anchor_rows = _row_built_to_reach_the_byte_bound()
    ix = _mk(x.shape[1], x)
    so, io = S.search_canonical(q, x, k)
    try:
        got = {}
        for permille in (1000, 750, 1000):
            _opt(b"debug_eps8_permille", permille)
            s, i = ix.index.search(q, k)
            st = ix.index.stats(reset=True)
            assert st["byte_scans"] == 1, (permille, st)
            got.setdefault(permille, []).append((s.copy(), i.copy()))
        for s, i in got[1000]:
            assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
        s, i = got[750][0]
        assert built not in i[0] and anchor_rows[0] in i[0], "a bound 25 % too small still found the built row: the construction has lost its teeth"
        assert sorted(set(io[0]) - set(i[0])) == [built]               # everything else is still right: the one row the bound exists for is the one that is lost
    finally:
        _opt(b"debug_eps8_permille", 1000)


def test_byte_prescan_rows_added_later_and_anisotropic_rows(byte_everywhere):
    """The int8 copy is derived data: rows added after the first search extend it (same centre and axis weights), a growth beyond its capacity rebuilds
    it; rows with a common direction and a few large-variance axes (what the centre / weights exist for) stay exact and mark few rows."""
    import torch
    from kirag_amd.bench_support import CorpusDist
    from kirag_amd.retriever.index import FlatIPIndex
    d, k = 1024, 10
    dev = torch.device("cuda:0")
    cd = CorpusDist("e5like", d, dev)
    g = torch.Generator(device=dev); g.manual_seed(5)
    x0 = cd.rows(60_000, g)
    ix = FlatIPIndex(d, device=0); ix.add(x0)
    q = cd.queries_near(x0[:3], g)
    s, i = ix.search(q, k)
    st = ix.stats(reset=True)
    so, io = S.search_canonical(q.cpu().numpy(), x0.cpu().numpy(), k)
    assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    assert st["byte_scans"] == 1 and st["byte_marked_rows"] < 60_000 // 8, st
    # a small add (fits the copy's padding) and a large one (rebuild), each with new best rows for query 0
    xs = [x0]
    for it, m in enumerate((100, 90_000)):
        extra = cd.rows(m, g)
        extra[:5] = torch.nn.functional.normalize(q[0] + 0.01 * torch.randn(5, d, device=dev, generator=g) / d ** 0.5, dim=1)
        n_before = sum(len(t) for t in xs)
        ix.add(extra); xs.append(extra)
        s, i = ix.search(q, k)
        st = ix.stats(reset=True)
        xa = torch.cat(xs).cpu().numpy()
        so, io = S.search_canonical(q.cpu().numpy(), xa, k)
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
        assert st["byte_scans"] == 1 and (i[0, : 5 * (it + 1)] >= 60_000).all() and (i[0] >= n_before).sum() == 5, (st, i[0])


def test_byte_prescan_through_a_streamed_build_of_small_appends(byte_everywhere):
    """A streamed build (cal_doc_embeddings(..., indexer=...): Indexer.index_data per batch, each followed by kr_index_prepare) interleaved with small searches: the
    int8 copy, the row bitmap (allocated for the index's CAPACITY, not re-allocated per append) and the list-length word behind the bits (which moves with the row count
    and must not leave a stale count inside the bitmap) follow 25 appends of odd sizes; every search equals the C oracle bit for bit and takes the pre-scan."""
    rng = np.random.default_rng(4242)
    d, k = 512, 7
    x = _unit(rng, 30_000 + 25 * 61, d)
    n = 30_000
    from kirag_amd.retriever.index import Indexer
    ix = Indexer(d)
    ix.index.reserve(len(x))
    ix.index_data([str(i) for i in range(n)], x[:n])
    ix.index.prepare(2, k)
    for step in range(25):
        q, pick = _queries_near(rng, x[:n], 2)
        s, i = ix.index.search(q, k)
        st = ix.index.stats(reset=True)
        so, io = S.search_canonical(q, x[:n], k)
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32)), step
        assert st["byte_scans"] == 1 and st["byte_rows"] == n and st["byte_marked_rows"] < n // 4, (step, st)
        m = 61 if step % 3 else 29                                   # odd sizes: the bitmap's word count changes on some appends and not on others
        ix.index_data([str(i) for i in range(n, n + m)], x[n:n + m]); n += m
        ix.index.prepare(2, k)
        assert ix.index.stats()["byte_rows"] == n


def test_byte_prescan_nan_row_marks_everything_once_then_steps_aside(byte_everywhere):
    """A non-finite element anywhere makes the byte bound meaningless: the pre-scan of that call marks every row (exact, slow), the index never takes
    the path again; NaN rows are never returned either way."""
    rng = np.random.default_rng(77)
    n, d, nq, k = 30000, 1024, 2, 10
    x = _unit(rng, n, d)
    x[12345] = np.nan
    q, _ = _queries_near(rng, x[:1000], nq)
    ix = _mk(d, x)
    xo = x.copy(); xo[12345] = 0.0
    so, io = S.search_canonical(q, xo, k)
    for call in range(2):
        s, i = ix.index.search(q, k)
        st = ix.index.stats(reset=True)
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32)) and not (i == 12345).any()
        assert st["byte_scans"] == (1 if call == 0 else 0), (call, st)
    

def test_byte_prescan_feedback_pause_and_resume_state_machine(byte_everywhere):
    """The feedback that keeps the pre-scan off data it cannot separate (search.hip: byte_bad / byte_pause).  A quarter of the rows are near-duplicates of one
    direction: a block aiming at that cluster has all of them inside the byte bound (marks > n / 8 rows), a block aiming anywhere else marks a handful.
    Three bad blocks, one good one (the count starts over), three bad ones (still scanning), the FOURTH bad block in a row pauses the path for the next 1024 small
    blocks (16-bit final round), block 1029 takes the pre-scan again, and four more bad blocks pause it again.  Results are the C oracle's through every
    transition."""
    rng = np.random.default_rng(1707)
    n, d, k = 40_000, 1024, 5
    x = _unit(rng, n, d)
    b = x[0].copy()
    z = rng.standard_normal((10_000, d)).astype(np.float32)
    x[:10_000] = b + 3e-5 * z
    x[:10_000] /= np.linalg.norm(x[:10_000], axis=1, keepdims=True)
    bad = (b + 1e-3 * rng.standard_normal((2, d)).astype(np.float32)); bad /= np.linalg.norm(bad, axis=1, keepdims=True)
    good, _ = _queries_near(rng, x[20_000:20_100], 2)
    good -= (good @ b)[:, None] * b[None, :]                  # orthogonal to the cluster's direction: its 10 000 rows score ~0 for these queries
    good = (good / np.linalg.norm(good, axis=1, keepdims=True)).astype(np.float32)
    ix = _mk(d, x)
    ref = {id(bad): S.search_canonical(bad, x, k), id(good): S.search_canonical(good, x, k)}

    def one(q, scan, many=None):
        s, i = ix.index.search(q, k)
        st = ix.index.stats(reset=True)
        so, io = ref[id(q)]
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
        assert st["byte_scans"] == (1 if scan else 0), st
        if many is not None:
            assert (st["byte_marked_rows"] > n // 8) == many, st
    for _ in range(3):
        one(bad, True, many=True)
    one(good, True, many=False)            # resets the count
    for _ in range(3):
        one(bad, True, many=True)
    one(bad, True, many=True)              # the fourth in a row: paused from here on
    for c in range(1024):
        one(bad if c % 2 else good, False)
    one(bad, True, many=True)              # block 1029: resumed
    for _ in range(3):
        one(bad, True, many=True)
    one(good, False)                       # ... and paused again after four more bad blocks


def test_index_prepare_builds_the_byte_copy_ahead_of_the_first_hop_and_gives_it_up_for_rows():
    """kr_index_prepare (ABI 9; Indexer.index_data / deserialize_from call it from 2^19 rows on): the int8 copy and the workspaces of a one-query top-10 search exist
    BEFORE the first hop (byte_rows == ntotal, first search = byte scan), rows added later extend the copy at the next prepare, searches on a torch side stream
    see a complete copy, results equal the unprepared index's bit for bit; an index below the threshold, a 33-query shape or byte_prescan = 0 build nothing."""
    import torch
    from kirag_amd.retriever.index import FlatIPIndex, Indexer
    n, d = 600_000, 1024
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(21)
    x = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
    q = torch.nn.functional.normalize(x[:2] + 0.05 * torch.randn(2, d, device=dev, generator=g), dim=1)
    plain = FlatIPIndex(d, device=0); plain.add(x)
    assert plain.stats()["byte_rows"] == 0
    s_ref, i_ref = plain.search(q, 10)
    ixr = Indexer(d)
    ixr.index_data([str(v) for v in range(400_000)], x[:400_000])
    assert ixr.index.stats()["byte_rows"] == 0                       # below 2^19 rows: nothing prepared
    ixr.index_data([str(v) for v in range(400_000, n)], x[400_000:])
    st = ixr.index.stats()
    assert st["byte_rows"] == n and st["byte_scans"] == 0, st        # built by index_data, no search yet
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                                    # a search on another stream right behind the build
        s1, i1 = ixr.index.search(q, 10)
    st = ixr.index.stats()
    assert st["byte_scans"] == 1 and np.array_equal(i1, i_ref) and np.array_equal(s1.view(np.uint32), s_ref.view(np.uint32)), st
    extra = torch.nn.functional.normalize(q[:1] + 0.01 * torch.randn(3, d, device=dev, generator=g), dim=1)
    ixr.index_data(["900001", "900002", "900003"], extra)            # the copy follows the rows
    assert ixr.index.stats()["byte_rows"] == n + 3
    res = ixr.search_knn(q[:1].cpu().numpy(), 5, verbose=False)
    assert set(res[0][0][:4]) == {"0", "900001", "900002", "900003"}
    # shapes / settings that do not take the pre-scan prepare workspaces only
    other = FlatIPIndex(d, device=0); other.add(x)
    other.prepare(33, 10)
    assert other.stats()["byte_rows"] == 0
    _opt(b"byte_prescan", 0)
    try:
        other.prepare(1, 10)
        assert other.stats()["byte_rows"] == 0
    finally:
        _opt(b"byte_prescan", 1)
    other.prepare(1, 10)
    assert other.stats()["byte_rows"] == n
    FlatIPIndex(d, device=0).prepare(1, 10)                          # an empty index: no-op


def test_byte_prescan_at_the_product_threshold_600k_rows():
    """The product's own switch-over (>= 2^19 rows): 600k x 1024 rows, 1 / 2 / 8 / 32 queries, top-10 and top-100, option on == option off bit for bit, a query
    bit-exact vs the C oracle, and the path is not taken by a 33-query block or below the threshold."""
    import torch
    from kirag_amd.retriever.index import FlatIPIndex
    n, d = 600_000, 1024
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(9)
    x = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
    ix = FlatIPIndex(d, device=0); ix.add(x)
    q = torch.nn.functional.normalize(x[:33] + 0.05 * torch.randn(33, d, device=dev, generator=g), dim=1)
    _opt(b"byte_prescan", 1); _opt(b"debug_byte_min_rows", -1)
    try:
        for nq in (1, 2, 8, 32):
            for k in (10, 100):
                s1, i1 = ix.search(q[:nq], k); st1 = ix.stats(reset=True)
                _opt(b"byte_prescan", 0)
                s0, i0 = ix.search(q[:nq], k); st0 = ix.stats(reset=True)
                _opt(b"byte_prescan", 1)
                assert st1["byte_scans"] == 1 and st0["byte_scans"] == 0 and st1["byte_marked_rows"] < (n // 4 if nq <= 8 else n), (st1, st0)
                assert np.array_equal(i1, i0) and np.array_equal(s1.view(np.uint32), s0.view(np.uint32)) and (i1[:, 0] == np.arange(nq)).all()
        s9, _ = ix.search(q, 10)
        assert ix.stats(reset=True)["byte_scans"] == 0
        so, io = S.search_canonical(q[:1].cpu().numpy(), x.cpu().numpy(), 10)
        s1, i1 = ix.search(q[:1], 10)
        assert np.array_equal(i1, io) and np.array_equal(s1.view(np.uint32), so.view(np.uint32))
        small = FlatIPIndex(d, device=0); small.add(x[:400_000])
        small.search(q[:1], 10)
        assert small.stats()["byte_scans"] == 0
    finally:
        _opt(b"byte_prescan", 1)


def test_faiss_padding_flag_for_k_above_ntotal():
    """`top_docs > ntotal`: default = ValueError (documented deviation); with faiss_padding=True the reference's behaviour on faiss's padded output:
    k entries per query, the missing ones with score -FLT_MAX and the id that label -1 maps to through index_id_to_db_id[-1] (index.py:49)."""
    from kirag_amd.retriever.index import Indexer
    rng = np.random.default_rng(3)
    x = _unit(rng, 7, 64); q = _unit(rng, 3, 64)
    ids = [str(100 + i) for i in range(7)]
    ix = Indexer(64); ix.index_data(ids, x)
    with pytest.raises(ValueError):
        ix.search_knn(q, 10, verbose=False)
    ixp = Indexer(64, faiss_padding=True); ixp.index_data(ids, x)
    out = ixp.search_knn(q, 10, verbose=False)
    so, io = S.search_canonical(q, x, 7)
    for r, (got_ids, got_s) in enumerate(out):
        assert got_ids[:7] == [str(100 + j) for j in io[r]] and got_ids[7:] == ["106"] * 3
        assert np.array_equal(got_s[:7], so[r]) and (got_s[7:] == -np.finfo(np.float32).max).all()
