"""world_size-2 gloo tests (CPU) of the N>1 path: the sharded search's all-gather + host merge equals the unsharded
search, and the in-batch gather helpers reproduce utils/utils.py:158-188."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import search_np as S
        from kirag_amd.parallel import ShardedSearcher
        from kirag_amd import utils as U
        from kirag_amd.compute_corpus_embeddings import shard_range
        rng = np.random.default_rng(0)
        x = rng.standard_normal((1001, 32)).astype(np.float32); q = rng.standard_normal((9, 32)).astype(np.float32)
        x[900] = x[3]                                            # a tie across the shard boundary
        a, b = shard_range(len(x), rank, world)

        class Shard:                                             # oracle-backed local index (CPU stand-in for FlatIPIndex)
            ntotal = b - a
            def search(self, qq, k): return S.search_canonical(np.asarray(qq), x[a:b], k)
        s, i = ShardedSearcher(Shard(), row_offset=a, world=world).search(torch.from_numpy(q), 20)
        so, io = S.search_canonical(q, x, 20)
        assert np.array_equal(i, io) and np.array_equal(s, so)
        # a shard smaller than k contributes what it has
        class Small:
            ntotal = 5 if rank == 0 else 600
            def search(self, qq, k): return S.search_canonical(np.asarray(qq), (x[:5] if rank == 0 else x[5:605]), k)
        s2, i2 = ShardedSearcher(Small(), row_offset=0 if rank == 0 else 5, world=world).search(torch.from_numpy(q), 8)
        so2, io2 = S.search_canonical(q, x[:605], 8)
        assert np.array_equal(i2, io2) and np.array_equal(s2, so2)
        # ShardedIndexer.search_knn: the "did any rank append rows?" decision is collective but rides behind the first batch's search (no round trip of
        # its own); a rank with an EMPTY share (never dirty itself) must still take part, and a later search must not assemble again
        from kirag_amd.retriever.index import ShardedIndexer

        class Rows:                                              # oracle-backed stand-in for FlatIPIndex (host queries, gloo: the CPU exchange path)
            def __init__(self): self.x = np.empty((0, 32), np.float32); self.device = 0
            @property
            def ntotal(self): return len(self.x)
            def add(self, e): self.x = np.concatenate([self.x, np.asarray(e, np.float32)])
            def search(self, qq, k): return S.search_canonical(np.asarray(qq), self.x, k)
        sx = ShardedIndexer.__new__(ShardedIndexer)
        sx.group, sx.rank, sx.world, sx.faiss_padding = None, rank, world, False
        sx.index, sx.index_id_to_db_id = Rows(), np.empty((0), dtype=np.int64)
        sx.row_offset, sx.ntotal_global, sx._local_ids, sx._dirty = 0, 0, [], False
        if rank == 1:                                            # rank 0's share of the streamed build is empty
            sx.index_data([str(7 * j + 1) for j in range(len(x))], x)
        calls = []
        orig = sx._assemble
        sx._assemble = lambda: (calls.append(1), orig())[1]
        res = sx.search_knn(q, 20, index_batch_size=4)          # three batches: the decision is taken behind the first one
        assert len(calls) == 1 and sx.ntotal_global == len(x) and sx.row_offset == 0
        for j, (ids, sc) in enumerate(res):
            assert ids == [str(7 * int(r) + 1) for r in io[j]] and np.array_equal(np.asarray(sc), so[j])
        res2 = sx.search_knn(q[:3], 5)                           # nothing appended since: no second assembly, optimistic search kept
        assert len(calls) == 1 and [r[0] for r in res2] == [[str(7 * int(r) + 1) for r in io[j][:5]] for j in range(3)]
        with pytest.raises(ValueError):
            sx.search_knn(q[:2], len(x) + 1)
        # in-batch helpers
        emb = torch.full((2, 4), float(rank), requires_grad=True)
        g = U.get_global_embeddings_for_inbatchtraining(rank, world, emb)
        assert tuple(g.shape) == (4, 4) and g.requires_grad and torch.equal(g[2 * rank:2 * rank + 2].detach(), emb.detach())
        assert torch.equal(g[:, 0].detach(), torch.tensor([0., 0., 1., 1.]))
        lab = U.get_global_labels_for_inbatchtraining(rank, world, torch.tensor([0, 1]), local_doc_size=3)
        assert lab.tolist() == [0, 1, 3, 4]
        assert U.get_global_labels_for_inbatchtraining(rank, world, None, 3) is None
        ret[rank] = "ok"
    except Exception as e:  # surface the failure in the parent
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_sharded_search_and_gathers_world2_gloo():
    from oracle import search_np as S
    S.build()                                                    # compile the C oracle once, before forking readers
    port = _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) == "ok" and ret.get(1) == "ok", (ret.get(0), ret.get(1))


def _worker4(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import search_np as S
        from kirag_amd.parallel import ShardedSearcher
        from kirag_amd.retriever.index import ShardedIndexer
        rng = np.random.default_rng(4)
        n, d, k = 700, 24, 16
        x = rng.standard_normal((n, d)).astype(np.float32); q = rng.standard_normal((11, d)).astype(np.float32)
        x[650] = x[2]; x[13] = x[2]                                # ties across three shards
        cuts = [0, 9, 300, 310, n]                                 # uneven: shards of 9 (< k), 291, 10 (< k) and 390 rows
        a, b = cuts[rank], cuts[rank + 1]

        class Shard:
            ntotal = b - a
            def search(self, qq, kk): return S.search_canonical(np.asarray(qq), x[a:b], kk)
        so, io = S.search_canonical(q, x, k)
        for exchange_first in (True, False):                        # the searcher's setting must not matter where the enqueue-only path cannot run
            sr = ShardedSearcher(Shard(), row_offset=a, world=world, exchange_first=exchange_first)
            s, i = sr.search(torch.from_numpy(q), k)
            assert np.array_equal(i, io) and np.array_equal(s, so), exchange_first
            with pytest.raises(ValueError):                         # host queries / a shard below k: the deferred form refuses, on EVERY rank alike
                sr.search_deferred(torch.from_numpy(q), k)
        # k beyond the whole corpus' smallest shards but within the corpus; and k > corpus: padded tail, the same on every rank
        s2, i2 = ShardedSearcher(Shard(), row_offset=a, world=world).search(q, n)
        so2, io2 = S.search_canonical(q, x, n)
        assert np.array_equal(i2, io2) and np.array_equal(s2, so2)
        s3, i3 = ShardedSearcher(Shard(), row_offset=a, world=world).search(q, n + 5)
        assert np.array_equal(i3[:, :n], io2) and (i3[:, n:] == -1).all() and np.isneginf(s3[:, n:]).all()
        # ShardedIndexer over the same uneven shards: the deferred / blocking decision comes from the smallest shard, which every rank knows
        class Rows:
            def __init__(self): self.x = np.empty((0, d), np.float32); self.device = 0
            @property
            def ntotal(self): return len(self.x)
            def add(self, e): self.x = np.concatenate([self.x, np.asarray(e, np.float32)])
            def search(self, qq, kk): return S.search_canonical(np.asarray(qq), self.x, kk)
            def prepare(self, *a_, **k_): pass
        sx = ShardedIndexer.__new__(ShardedIndexer)
        sx.group, sx.rank, sx.world, sx.faiss_padding = None, rank, world, False
        sx.index, sx.index_id_to_db_id = Rows(), np.empty((0), dtype=np.int64)
        sx.row_offset, sx.ntotal_global, sx._local_ids, sx._dirty, sx._min_shard_rows, sx.deferred_blocks = 0, 0, [], False, None, 0
        sx.index_data([str(5 * j + 3) for j in range(a, b)], x[a:b])
        res = sx.search_knn(q, k, index_batch_size=4)
        assert sx._min_shard_rows == 9 and sx.row_offset == a and sx.ntotal_global == n and not sx._deferred_ok(k) and sx.deferred_blocks == 0
        for j, (ids, sc) in enumerate(res):
            assert ids == [str(5 * int(r) + 3) for r in io[j]] and np.array_equal(np.asarray(sc), so[j])
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_sharded_search_world4_uneven_shards_below_k_gloo():
    """four ranks, shards of 9 / 291 / 10 / 390 rows with k = 16 (two shards hold fewer than k rows), searcher configured with
    exchange_first=True and False: the blocking search is the unsharded canonical answer on every rank, the enqueue-only form refuses on every rank alike, k up to
    and beyond the corpus size pads identically, and ShardedIndexer takes the deferred / blocking decision from the smallest shard (known to all ranks)."""
    from oracle import search_np as S
    S.build()
    port = _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker4, args=(4, port, ret), nprocs=4, join=True)
    assert all(ret.get(r) == "ok" for r in range(4)), [ret.get(r) for r in range(4)]
