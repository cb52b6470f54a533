"""``ShardedIndexer``: the reference's ``Indexer`` surface (``retriever/index.py:17-83``) over a corpus whose rows are split contiguously over the ranks of a
``torch.distributed`` group (BASELINE config 3 / SURVEY.md 8e) — local exact top-k per shard, one all-gather of the per-shard lists, merge."""
from __future__ import annotations

import logging
import os
import pickle

import numpy as np

from .flat_index import ids_to_str_rows
from .formats import (SHARD_MANIFEST, _ids_crc32, _manifest_matches, read_faiss_flat_ip, read_native_shards, shard_file_name, write_native_shard)
from .index import Indexer

logger = logging.getLogger()


class ShardedIndexer(Indexer):
    """``Indexer`` whose rows are split contiguously over the ranks of a ``torch.distributed`` group — BASELINE config 3 (5M-passage corpus
    row-sharded over the 8 GPUs of a node, RCCL all-gather of per-shard top-k) behind the reference's own ``Indexer`` surface.

    Rank r holds rows ``[r * ceil(N / W), (r + 1) * ceil(N / W))`` of the index in its own HBM and the FULL ``index_id_to_db_id`` map
    (8 bytes per row).  ``search_knn`` is collective: every rank passes the same queries and gets the global top-k — local exact top-k,
    ONE all-gather of ``nq * k * 12`` bytes per rank, host-side merge by (score desc, global row asc) — identical to the unsharded
    ``Indexer`` (tested).  ``index.ntotal`` is the LOCAL row count; ``ntotal_global`` the corpus size."""

    def __init__(self, vector_sz, metric="inner_product", n_subquantizers=0, n_bits=8, device=None, coarse_dtype="bf16", group=None):
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        super().__init__(vector_sz, metric=metric, n_subquantizers=n_subquantizers, n_bits=n_bits, device=device, coarse_dtype=coarse_dtype)
        self.row_offset = 0
        self.ntotal_global = 0
        self._local_ids = []
        self._dirty = False
        self._min_shard_rows = None      # rows of the smallest shard, known to EVERY rank (None: not known — the deferred search path stays off)
        self.deferred_blocks = 0         # query blocks answered through the enqueue-only path (tests)

    def index_data(self, ids, embeddings):
        """Streamed build (``cal_doc_embeddings(..., indexer=this)`` on every rank, BASELINE config 4: "streamed encode + search"): appends this
        rank's rows to its resident shard — device tensors go in device-to-device, stream-ordered — and remembers their ids.  Not collective;
        the global id map and the row offsets are exchanged by ``sync_shards()``, which ``search_knn`` / ``serialize`` call when needed."""
        if isinstance(embeddings, np.ndarray):
            embeddings = embeddings.astype('float32')
        self.index.add(embeddings)
        self._local_ids.append(np.array(ids, dtype=np.int64))
        self._dirty = True
        self._min_shard_rows = None
        self._prepare_small_searches()

    def _dirty_flag(self, async_op: bool):
        """the collective part of the "did any rank append rows?" decision: (flag tensor, work handle or None).  A rank whose share of a streamed build
        was empty (or that appended nothing after a reload) has _dirty == False while the others do: the decision has to be taken together."""
        import torch
        import torch.distributed as dist
        backend = dist.get_backend(self.group)
        flag = torch.tensor([1 if self._dirty else 0], dtype=torch.int32,
                            device=torch.device("cuda", self.index.device) if backend == "nccl" else torch.device("cpu"))
        work = dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group, async_op=async_op)
        return flag, work

    def _assemble(self):
        """Collective: ``index_id_to_db_id`` (all ranks' ids in rank order), ``row_offset`` and ``ntotal_global`` from every rank's local ids."""
        import torch.distributed as dist
        local = np.concatenate(self._local_ids, axis=0) if self._local_ids else np.empty((0), dtype=np.int64)
        parts = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(parts, local, group=self.group)
        else:
            parts = [local]
        self.row_offset = int(sum(len(p) for p in parts[: self.rank]))
        self.index_id_to_db_id = np.concatenate(parts, axis=0)
        self.ntotal_global = len(self.index_id_to_db_id)
        self._local_ids = [local]
        self._dirty = False
        self._min_shard_rows = min(len(p) for p in parts)

    def sync_shards(self):
        """Collective: assemble ``index_id_to_db_id``, ``row_offset`` and ``ntotal_global`` after ``index_data`` calls (no-op when no rank appended rows)."""
        dirty = bool(self._dirty)
        if self.world > 1:
            flag, _ = self._dirty_flag(async_op=False)
            dirty = bool(int(flag.item()))
        if dirty:
            self._assemble()

    def set_local_shard(self, local_ids, embeddings):
        """Resident-shard build path (``compute_corpus_embeddings.cal_doc_embeddings(..., indexer=...)`` on every rank): this rank contributes
        its contiguous rows; the global id map and the row offsets are assembled with one ``all_gather_object``.  Collective."""
        import torch.distributed as dist
        if isinstance(embeddings, np.ndarray):
            embeddings = embeddings.astype('float32')
        self.index.add(embeddings)
        local = np.array(local_ids, dtype=np.int64)
        parts = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(parts, local, group=self.group)
        else:
            parts = [local]
        self.row_offset = int(sum(len(p) for p in parts[: self.rank]))
        self.index_id_to_db_id = np.concatenate(parts, axis=0)
        self.ntotal_global = len(self.index_id_to_db_id)
        self._local_ids, self._dirty = [local], False
        self._min_shard_rows = min(len(p) for p in parts)
        self._prepare_small_searches()

    def deserialize_from(self, dir_path):
        """Loads this rank's contiguous share of the rows.  Preferred source: the native shard files written by ``serialize`` (fp32 rows + the 16-bit
        scan copy + its error bounds: no re-quantisation; any saved world size can be loaded into any other).  Otherwise the reference's
        ``index.faiss`` (``index.py:66-79``), of which each rank reads only its byte range."""
        meta_file = os.path.join(dir_path, "index_meta.faiss")
        manifest = os.path.join(dir_path, SHARD_MANIFEST)
        with open(meta_file, "rb") as reader:
            id_map = pickle.load(reader)
        if os.path.exists(manifest) and not _manifest_matches(manifest, id_map):
            logger.warning(f'{manifest} does not belong to {meta_file} (row count or id-map checksum differ): ignoring the native shards')
            manifest = None
        if manifest is not None and os.path.exists(manifest):
            logger.info(f'Loading rank {self.rank}/{self.world} share of the native shards in {dir_path}')
            self.index = read_native_shards(dir_path, device=self.index.device, coarse_dtype=self.index.coarse_dtype, row_range=(self.rank, self.world))
        else:
            index_file = os.path.join(dir_path, "index.faiss")
            logger.info(f'Loading rank {self.rank}/{self.world} share of {index_file}, meta data from {meta_file}')
            self.index = read_faiss_flat_ip(index_file, device=self.index.device, coarse_dtype=self.index.coarse_dtype, row_range=(self.rank, self.world))
        self.index_id_to_db_id = id_map
        self.row_offset, self.ntotal_global = self.index.row_offset, self.index.file_ntotal
        assert len(self.index_id_to_db_id) == self.ntotal_global, 'Deserialized index_id_to_db_id should match faiss index size'
        # later index_data() calls append to this rank's shard: its ids so far are its slice of the loaded map
        self._local_ids = [np.asarray(self.index_id_to_db_id[self.row_offset: self.row_offset + self.index.ntotal], dtype=np.int64)]
        self._dirty = False
        per = (self.ntotal_global + self.world - 1) // self.world          # the contiguous split both readers use (row_range = (rank, world))
        self._min_shard_rows = min(max(0, min((r + 1) * per, self.ntotal_global) - min(r * per, self.ntotal_global)) for r in range(self.world))
        self._prepare_small_searches()

    def serialize(self, dir_path):
        """Collective.  Every rank writes its resident rows as one native shard file (``index_shard_RRRR_of_WWWW.krshard``); rank 0 also writes the
        manifest (``kirag_shards.json``) and ``index_meta.faiss`` — the reference's pickled int64 id map (``index.py:63-64``), so the meta file is
        the one every other tool of the reference expects.  ``index.faiss`` itself is NOT written (a 5M x 1024 index is 20 GB that would have to
        funnel through one rank); ``faiss_index_corpus`` / ``Indexer.serialize`` produce it when a single-host index is wanted."""
        import torch.distributed as dist
        self.sync_shards()
        os.makedirs(dir_path, exist_ok=True)
        n_local = self.index.ntotal
        write_native_shard(self.index, os.path.join(dir_path, shard_file_name(self.rank, self.world)), self.row_offset, self.ntotal_global)
        info = [None] * self.world
        mine = {"rank": self.rank, "row0": int(self.row_offset), "rows": int(n_local), "file": shard_file_name(self.rank, self.world)}
        if self.world > 1:
            dist.all_gather_object(info, mine, group=self.group)
        else:
            info = [mine]
        if self.rank == 0:
            import json
            with open(os.path.join(dir_path, SHARD_MANIFEST), "w") as f:
                json.dump({"format": "krshard-1", "meta_crc32": _ids_crc32(self.index_id_to_db_id), "d": self.index.d, "coarse_dim": self.index.coarse_dim, "coarse_dtype": self.index.coarse_dtype,
                           "ntotal": int(self.ntotal_global), "world": self.world, "shards": sorted(info, key=lambda e: e["row0"])}, f, indent=1)
            with open(os.path.join(dir_path, "index_meta.faiss"), mode='wb') as f:
                pickle.dump(self.index_id_to_db_id, f)
        if self.world > 1:
            dist.barrier(group=self.group)

    def _deferred_ok(self, k: int) -> bool:
        """Collective-safe: depends only on values every rank holds (world, k, the smallest shard's row count, the backend)."""
        import torch
        import torch.distributed as dist
        from ..parallel import ShardedSearcher
        return (self.world > 1 and dist.is_initialized() and torch.cuda.is_available() and self._min_shard_rows is not None
                and 0 < k <= self._min_shard_rows and self.world * k <= ShardedSearcher.DEVICE_MERGE_MAX)

    def _search_knn_deferred(self, query_vectors, starts, bs, k, result):
        import torch
        sr = self._get_searcher()
        dev = torch.device("cuda", self.index.device)
        lo = starts[0]
        qd = query_vectors[lo:] if torch.is_tensor(query_vectors) else torch.from_numpy(np.ascontiguousarray(query_vectors[lo:], dtype=np.float32))
        qd = qd.detach().to(dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            sr.search_deferred(qd[0: bs], k)
            for j, s0 in enumerate(starts):
                ps, pi = sr.finish_deferred()[-1]                      # block j final (the only search outstanding)
                scores, rows = ps.numpy().copy(), pi.numpy().copy()
                if j + 1 < len(starts):
                    a = starts[j + 1] - lo
                    sr.search_deferred(qd[a: a + bs], k)
                db_ids = ids_to_str_rows(self.index_id_to_db_id[rows])
                result.extend([(db_ids[i], scores[i]) for i in range(len(db_ids))])
                self.deferred_blocks += 1

    def _get_searcher(self):
        from ..parallel import ShardedSearcher
        sr = getattr(self, "_searcher", None)
        if sr is None or sr.row_offset != self.row_offset or sr.index is not self.index:
            sr = self._searcher = ShardedSearcher(self.index, row_offset=self.row_offset, world=self.world, group=self.group)
        return sr

    def search_knn(self, query_vectors, top_docs: int, index_batch_size=1024, verbose: bool = True):
        """Collective (every rank passes the same queries).  Whether some rank appended rows since the last ``sync_shards()`` is decided together, but
        NOT with a host round trip of its own per call: the flag's all-reduce is started before the first batch's search and looked at behind that
        search's own synchronisation; only if it says "dirty" (the first search after a streamed build) is the id map assembled and that batch
        searched again with the right row offsets.  (Every branch below depends on state that is identical on all ranks, so the ranks issue the same
        collectives in the same order.)"""
        if isinstance(query_vectors, np.ndarray):
            query_vectors = query_vectors.astype('float32')
        top_docs = int(top_docs)
        flag = work = None
        if self.world > 1:
            flag, work = self._dirty_flag(async_op=True)
        elif self._dirty:
            self._assemble()
        result = []
        starts = list(range(0, len(query_vectors), index_batch_size))
        done = 0
        for start_idx in starts:
            q = query_vectors[start_idx: start_idx + index_batch_size]
            res = None
            if flag is not None:
                if 0 < top_docs <= self.ntotal_global:         # optimistic: the common case is "nobody appended anything"
                    res = self._get_searcher().search(q, top_docs)
                work.wait()
                if bool(int(flag.item())):                     # behind the search's synchronisation: no round trip of its own
                    self._assemble()
                    res = None                                 # searched with stale row offsets / id map: again
                flag = None
            if not 0 < top_docs <= self.ntotal_global:
                raise ValueError(f"top_docs={top_docs} must satisfy 0 < k <= ntotal={self.ntotal_global}")
            scores, rows = res if res is not None else self._get_searcher().search(q, top_docs)
            db_ids = ids_to_str_rows(self.index_id_to_db_id[rows])
            result.extend([(db_ids[i], scores[i]) for i in range(len(db_ids))])
            done += 1
            if done < len(starts) and self._deferred_ok(top_docs):
                # the remaining blocks, pipelined like Indexer.search_knn: block j + 1 (local search + all-gather + device merge + D2H, enqueue only) runs
                # on the device while the host builds block j's id strings.  Taken from state every rank shares, so the ranks branch alike.
                self._search_knn_deferred(query_vectors, starts[done:], index_batch_size, top_docs, result)
                break
        if flag is not None:                                   # no query batch at all: still take the collective decision
            work.wait()
            if bool(int(flag.item())):
                self._assemble()
        return result
