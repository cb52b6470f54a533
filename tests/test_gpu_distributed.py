"""world_size-2 test of the N>1 search path with REAL device shards: two processes share the one GPU of the test box (each holds half of
the corpus in a FlatIPIndex), queries live on the device, ShardedSearcher goes through search_into + all_gather + host merge.  The collective
backend is gloo here (RCCL refuses two ranks on one device); on a multi-GPU node the same code runs over backend "nccl" = RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import search_np as S
        from kirag_amd.compute_corpus_embeddings import shard_range
        from kirag_amd.parallel import ShardedSearcher
        from kirag_amd.retriever.index import FlatIPIndex
        rng = np.random.default_rng(0)
        n, d, nq, k = 20000, 256, 33, 100
        x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
        q = x[rng.choice(n, nq)] + 0.1 * rng.standard_normal((nq, d)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
        x[n // 2 + 7] = x[11]                                    # a tie across the shard boundary
        a, b = shard_range(n, rank, world)
        ix = FlatIPIndex(d, device=0); ix.add(torch.from_numpy(x[a:b]).cuda())
        s, i = ShardedSearcher(ix, row_offset=a, world=world).search(torch.from_numpy(q).cuda(), k)
        so, io = S.search_canonical(q, x, k)
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_sharded_search_world2_device_shards():
    from oracle import search_np as S
    S.build()
    port = _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) == "ok" and ret.get(1) == "ok", (ret.get(0), ret.get(1))
