"""Drop-in for the reference's ``retriever/index.py`` (``Indexer`` :17-83) with the flat inner-product index
living in MI355X HBM (``libkirag_amd.so``: MFMA coarse scan + certified exact fp64 re-rank) instead of
``faiss.IndexFlatIP`` on host cores.

Same surface: ``Indexer(vector_sz, metric="inner_product", n_subquantizers=0, n_bits=8)``,
``index_data(ids, embeddings)``, ``search_knn(query_vectors, top_docs, index_batch_size=1024, verbose=True)``
-> ``List[Tuple[List[str], np.ndarray(float32)[k]]]``, ``serialize(dir)``, ``deserialize_from(dir)``,
attributes ``index.ntotal`` and ``index_id_to_db_id`` (``np.int64``).

Differences, all deliberate and documented in DESIGN.md:
  * scores are the canonical score (fp64-accumulated, rounded once to fp32) and ties are broken by internal
    row number — faiss leaves both to its BLAS; results are deterministic here.
  * ``top_docs > ntotal`` raises ``ValueError`` (faiss pads with label -1, which ``index.py:49`` would silently
    map to the LAST id).
  * ``metric="l2"`` and ``n_subquantizers > 0`` (``IndexPQ``) raise ``NotImplementedError``: no caller in the
    reference uses them (``faiss_index_corpus.py:29``, ``retrieve.py:109`` pass ``metric="inner_product"``).
"""
from __future__ import annotations

import logging
import os
import pickle
from typing import List, Tuple

import numpy as np

from .flat_index import FlatIPIndex, ids_to_str_rows
from .formats import SHARD_MANIFEST, read_faiss_flat_ip, read_native_shards, write_faiss_flat_ip, write_native_shard   # noqa: F401  (re-exported: round 1-5 import paths)

logger = logging.getLogger()


class Indexer(object):

    def __init__(self, vector_sz, metric="inner_product", n_subquantizers=0, n_bits=8, device=None, coarse_dtype="bf16", faiss_padding=False):
        """``faiss_padding=True`` reproduces what the reference does when ``top_docs > ntotal`` instead of raising: faiss pads the result with
        label -1 and score -FLT_MAX, and ``index.py:49`` maps label -1 through ``index_id_to_db_id[-1]`` to the LAST id — recalled from faiss's
        documented behaviour (not verifiable here: faiss is absent), so it stays opt-in."""
        self.faiss_padding = bool(faiss_padding)
        if n_subquantizers > 0:
            raise NotImplementedError("IndexPQ (n_subquantizers > 0) is not part of the MI355X path; no reference caller uses it")
        if metric != "inner_product":
            raise NotImplementedError(f"metric={metric!r}: only 'inner_product' (IndexFlatIP) is implemented")
        self.index = FlatIPIndex(vector_sz, device=device, coarse_dtype=coarse_dtype)
        self.index_id_to_db_id = np.empty((0), dtype=np.int64)

    PREPARE_FROM_ROWS = 1 << 19     # index size from which small query blocks stream the int8 copy (kr_set_option "debug_byte_min_rows" default)

    def index_data(self, ids, embeddings):
        # reference order: id map first, then astype('float32'), then add (index.py:28-32)
        self._update_id_mapping(ids)
        if isinstance(embeddings, np.ndarray):
            embeddings = embeddings.astype('float32')
        self.index.add(embeddings)
        logger.info(f'Total data indexed {len(self.index_id_to_db_id)}')
        self._prepare_small_searches()

    def _prepare_small_searches(self):
        """First-touch costs out of the first hop: once the index is large enough for the byte pre-scan, the int8 copy of the rows is
        kept up to date as rows arrive (``kr_index_prepare``: quantises the NEW rows only, ~1.7 us per 1000 rows) together with the workspaces of a one-query
        top-10 search — the KiRAG loop's hop (knowledge_graph/models.py:1645).  The library decides whether the copy is affordable (free HBM) and drops it
        again if the rows themselves need the memory."""
        if self.index.ntotal >= self.PREPARE_FROM_ROWS:
            self.index.prepare(1, 10)

    accepts_device_queries = True      # search_knn takes a torch tensor on the index's GPU as it takes a numpy array (DenseRetriever.batch_retrieve)

    def search_knn(self, query_vectors, top_docs: int, index_batch_size=1024, verbose: bool = True) -> List[Tuple[List[object], List[float]]]:
        """index.py:36-53, pipelined: the reference's loop searches a block, THEN builds its id strings, THEN starts the next block — the string pass
        costs as much host time as the GPU's whole search of the block.  Here two blocks are in flight (``kr_index_search_async``, results into pinned
        host buffers; the oldest is finished with ``kr_index_search_finish_one``) while block i's ids are converted, so the device never waits for the
        host's conversion; the last block is searched as 3/4 + 1/4 so that only a quarter block's strings are built behind the device; same lists, same order."""
        if isinstance(query_vectors, np.ndarray):
            query_vectors = np.asarray(query_vectors, dtype=np.float32)        # (index.py:40 `.astype('float32')`, without the copy when it already is)
        top_docs = int(top_docs)
        nq_all = len(query_vectors)
        blocks = [(s, min(s + index_batch_size, nq_all)) for s in range(0, nq_all, index_batch_size)]
        result = []
        if self.faiss_padding and top_docs > self.index.ntotal:
            for s0, e0 in blocks:
                q = query_vectors[s0:e0]
                n = self.index.ntotal
                scores = np.full((len(q), top_docs), -np.finfo(np.float32).max, np.float32)
                indexes = np.full((len(q), top_docs), -1, np.int64)
                if n > 0:
                    s_, i_ = self.index.search(q, n)
                    scores[:, :n] = s_; indexes[:, :n] = i_
                db_ids = ids_to_str_rows(self.index_id_to_db_id[indexes])
                result.extend([(db_ids[i], scores[i]) for i in range(len(db_ids))])
            return result
        query_vectors = self._on_index_device(query_vectors)
        if len(blocks) <= 1:
            for s0, e0 in blocks:
                scores, indexes = self.index.search(query_vectors[s0:e0], top_docs)
                db_ids = ids_to_str_rows(self.index_id_to_db_id[indexes])   # external ids (vectorised form of index.py:49)
                result.extend([(db_ids[i], scores[i]) for i in range(len(db_ids))])
            return result
        import torch
        if not 0 < top_docs <= self.index.ntotal:
            raise ValueError(f"top_docs={top_docs} must satisfy 0 < k <= ntotal={self.index.ntotal}")
        # the conversion of the LAST block's ids has no search left to hide under: that block is searched as two pieces (3/4 + 1/4, whole 128-query tiles), so
        # only a quarter block's strings (~2 ms instead of ~8) are built after the device has finished; same lists, same order
        s_last, e_last = blocks[-1]
        if e_last - s_last >= 512:
            tail = max(128, ((e_last - s_last) // 4) // 128 * 128)
            blocks[-1:] = [(s_last, e_last - tail), (e_last - tail, e_last)]
        dev = torch.device("cuda", self.index.device)
        d = query_vectors.shape[1]
        upload = None
        if torch.is_tensor(query_vectors):
            qd = query_vectors.detach().to(dev, dtype=torch.float32).contiguous()
        else:
            # Host queries (what index.py:36 takes) go up BLOCK BY BLOCK, each just ahead of its own search, through a pinned staging buffer and a copy stream
            # kept between calls: only block 0's 4 MiB are uploaded before the first search starts; the memcpy into pinned memory and the DMA of block j + 2
            # run while the device searches block j (one 16-MiB upload up front cost 3.5 ms of a 45-ms call; from pageable memory it is a synchronous staged
            # copy, and uploads on the search stream would queue behind the searches in flight).
            cap = max(nq_all, 4096)
            if getattr(self, "_knn_q_stage", None) is None or self._knn_q_stage.shape[0] < nq_all or self._knn_q_stage.shape[1] != d:
                self._knn_q_stage = torch.empty((cap, d), dtype=torch.float32, pin_memory=True)
                self._knn_q_dev = torch.empty((cap, d), dtype=torch.float32, device=dev)
                self._knn_copy_stream = torch.cuda.Stream(device=dev)
            stage, qd, side = self._knn_q_stage, self._knn_q_dev, self._knn_copy_stream
            stage_np = stage.numpy()

            def upload(j):
                s0, e0 = blocks[j]
                stage_np[s0:e0] = query_vectors[s0:e0]
                with torch.cuda.stream(side):
                    qd[s0:e0].copy_(stage[s0:e0], non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(side)
                torch.cuda.current_stream(dev).wait_event(ev)          # the search of block j (enqueued next, on the current stream) starts behind its upload
        bs = blocks[0][1] - blocks[0][0]
        key = (bs, top_docs)
        if getattr(self, "_knn_slots_key", None) != key:   # the three pinned result slots are kept between calls (pinning 2.4 MiB costs more than a block's upload)
            self._knn_slots = [(torch.empty((bs, top_docs), dtype=torch.float32, pin_memory=True), torch.empty((bs, top_docs), dtype=torch.int64, pin_memory=True))
                               for _ in range(3)]
            self._knn_slots_key = key
        slots = self._knn_slots

        def enqueue(j):
            s0, e0 = blocks[j]
            ps, pi = slots[j % 3]
            if upload is not None:
                upload(j)
            self.index.search_async(qd[s0:e0], top_docs, ps[:e0 - s0], pi[:e0 - s0])

        # the cyclic garbage collector is paused for the call: a block creates 1024 lists of 100 strings each (+ 1024 tuples), every 700th container allocation
        # starts a collection and the older generations then traverse all of them again - 5 to 10 ms of a 42-ms call, more in a process with many live objects
        # (nothing here creates reference cycles; collection resumes when the call returns)
        import gc
        gc_was_on = gc.isenabled()
        with torch.cuda.device(dev):
            try:
                gc.disable()
                enqueue(0)
                if len(blocks) > 1:
                    enqueue(1)                                          # two calls in flight: the device goes from block j straight into block j + 1
                for j, (s0, e0) in enumerate(blocks):
                    self.index.finish_one()                             # block j (the oldest call) is final in its pinned slot
                    ps, pi = slots[j % 3]
                    scores = ps[:e0 - s0].numpy().copy(); indexes = pi[:e0 - s0].numpy().copy()
                    if j + 2 < len(blocks):
                        enqueue(j + 2)                                  # slot (j + 2) % 3 was read out one iteration ago
                    db_ids = ids_to_str_rows(self.index_id_to_db_id[indexes])
                    result.extend([(db_ids[i], scores[i]) for i in range(len(db_ids))])
            except BaseException:
                try:
                    self.index.finish()                                 # nothing of this call stays outstanding on the handle
                except Exception:
                    pass
                raise
            finally:
                if gc_was_on:
                    # Re-enabling alone makes the caller's next container allocation start a young-generation collection over everything this call created (4096 lists
                    # of 100 strings: 410 k item visits, a few ms that no timer around this call sees).  gc.freeze() + gc.unfreeze() (documented API, O(1) list
                    # splices) hand all currently tracked objects to the OLDEST generation without traversing them and reset the allocation counters: the result
                    # lists are examined at the process's next full collection, like any long-lived data.
                    if hasattr(gc, "freeze") and gc.get_freeze_count() == 0:
                        gc.freeze(); gc.unfreeze()
                    gc.enable()
        return result

    def _on_index_device(self, q):
        """A CUDA tensor of queries on ANOTHER GPU than the index's (the encoder on cuda:1, ``Indexer(device=None)`` on LOCAL_RANK / 0) is moved with ``.to()``:
        torch orders that cross-device copy behind the encoder's kernels on the source GPU's stream and before the search on this one's (the C
        side would otherwise stage the foreign pointer with a hipMemcpyAsync on the index GPU's stream, which nothing orders after the encoder)."""
        if not isinstance(q, np.ndarray) and getattr(q, "is_cuda", False) and q.device.index != self.index.device:
            import torch
            return q.to(torch.device("cuda", self.index.device))
        return q

    # ---- on-disk formats (index.py:55-79) --------------------------------------------------------------
    def serialize(self, dir_path):
        index_file = os.path.join(dir_path, "index.faiss")
        meta_file = os.path.join(dir_path, "index_meta.faiss")
        logger.info(f'Serializing index to {index_file}, meta data to {meta_file}')
        write_faiss_flat_ip(self.index, index_file)
        with open(meta_file, mode='wb') as f:
            pickle.dump(self.index_id_to_db_id, f)
        stale = os.path.join(dir_path, SHARD_MANIFEST)      # native shards of an EARLIER index in this directory must not shadow these files on reload
        if os.path.exists(stale):
            os.remove(stale)

    def deserialize_from(self, dir_path):
        index_file = os.path.join(dir_path, "index.faiss")
        meta_file = os.path.join(dir_path, "index_meta.faiss")
        logger.info(f'Loading index from {index_file}, meta data from {meta_file}')
        self.index = read_faiss_flat_ip(index_file, device=self.index.device, coarse_dtype=self.index.coarse_dtype)
        logger.info('Loaded index of type %s and size %d', type(self.index), self.index.ntotal)
        with open(meta_file, "rb") as reader:
            self.index_id_to_db_id = pickle.load(reader)
        assert len(
            self.index_id_to_db_id) == self.index.ntotal, 'Deserialized index_id_to_db_id should match faiss index size'
        self._prepare_small_searches()

    def _update_id_mapping(self, db_ids: List):
        new_ids = np.array(db_ids, dtype=np.int64)
        self.index_id_to_db_id = np.concatenate((self.index_id_to_db_id, new_ids), axis=0)


from .sharded import ShardedIndexer   # noqa: E402,F401  (lives in sharded.py; imported here so that `from kirag_amd.retriever.index import ShardedIndexer` keeps working)
