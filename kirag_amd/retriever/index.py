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

import ctypes as C
import logging
import os
import pickle
import struct
from typing import List, Optional, Tuple

import numpy as np

from .. import _lib

logger = logging.getLogger()

_IO_CHUNK_ROWS = 1 << 18


class FlatIPIndex:
    """The object behind ``Indexer.index`` — the subset of ``faiss.IndexFlatIP`` the reference touches
    (``ntotal``, ``d``, ``is_trained``, ``add``, ``search``) backed by a ``kr_index`` handle."""

    is_trained = True

    def __init__(self, d: int, device: Optional[int] = None, coarse_dtype: str = "bf16"):
        lib = _lib.load()
        if device is None:
            device = int(os.environ.get("KIRAG_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.d = int(d)
        self.device = int(device)
        self.coarse_dtype = coarse_dtype
        h = C.c_void_p()
        _lib.check(lib.kr_index_create(self.d, 0, {"bf16": 0, "f16": 1}[coarse_dtype], self.device, C.byref(h)))
        self._h = h
        self._lib = lib

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.kr_index_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def ntotal(self) -> int:
        return int(self._lib.kr_index_ntotal(self._h))

    def reserve(self, n_rows: int) -> None:
        _lib.check(self._lib.kr_index_reserve(self._h, int(n_rows)))

    @staticmethod
    def _ptr(a):
        """(pointer, keepalive) for a numpy array or a torch tensor (host or device)."""
        if isinstance(a, np.ndarray):
            return a.ctypes.data, a
        return int(a.data_ptr()), a  # torch.Tensor

    def _stream(self, *tensors):
        """hipStream_t for a call: torch's CURRENT stream on this index's device when any operand is a CUDA tensor (work the caller queued on a
        side stream — a cast, a gather — is then ordered before the library's kernels, and the outputs after them), else the default stream."""
        for t in tensors:
            if not isinstance(t, np.ndarray) and getattr(t, "is_cuda", False):
                import torch
                return int(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def add(self, x) -> None:
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, dtype=np.float32)
        else:
            x = x.detach().float().contiguous()
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n,{self.d}] embeddings, got {tuple(x.shape)}")
        p, keep = self._ptr(x)
        _lib.check(self._lib.kr_index_add(self._h, p, int(x.shape[0]), self._stream(x)))

    def search(self, q, k: int, mode: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """(scores float32 [nq,k] descending, internal rows int64 [nq,k]) — faiss's (D, I)."""
        if isinstance(q, np.ndarray):
            q = np.ascontiguousarray(q, dtype=np.float32)
        else:
            q = q.detach().float().contiguous()
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"expected [nq,{self.d}] queries, got {tuple(q.shape)}")
        k = int(k)
        if not 0 < k <= self.ntotal:
            raise ValueError(f"top_docs={k} must satisfy 0 < k <= ntotal={self.ntotal}")
        nq = int(q.shape[0])
        p, keep = self._ptr(q)
        pin = self._small_out(nq, k) if mode == 0 else None
        if pin is not None:
            # a small call (the KiRAG loop's 1-2 queries per hop): results land in a pinned scratch the device writes directly (ABI 8: the library works
            # in place when it can address the caller's buffers) instead of going through a staged copy into pageable memory
            ps, pr, pq = pin
            if isinstance(q, np.ndarray):                       # host queries: through the pinned query scratch (4 KiB per query) for the same reason
                pq[: nq * self.d].numpy()[:] = q.reshape(-1)
                p = int(pq.data_ptr())
            _lib.check(self._lib.kr_index_search(self._h, p, nq, k, int(ps.data_ptr()), int(pr.data_ptr()), 0, self._stream(q)))
            return ps[: nq * k].numpy().reshape(nq, k).copy(), pr[: nq * k].numpy().reshape(nq, k).copy()
        scores = np.empty((nq, k), np.float32)
        rows = np.empty((nq, k), np.int64)
        _lib.check(self._lib.kr_index_search(self._h, p, nq, k, scores.ctypes.data, rows.ctypes.data, int(mode), self._stream(q)))
        return scores, rows

    def _small_out(self, nq: int, k: int):
        """pinned result scratch for calls of at most 32 queries x 1024 hits (None: larger call, or no torch / no GPU to pin for)"""
        if nq > 32 or nq * k > 32 * 1024:
            return None
        pin = getattr(self, "_pin_out", None)
        if pin is None:
            try:
                import torch
                if not torch.cuda.is_available():
                    raise RuntimeError
                pin = (torch.empty(32 * 1024, dtype=torch.float32, pin_memory=True), torch.empty(32 * 1024, dtype=torch.int64, pin_memory=True),
                       torch.empty(32 * self.d, dtype=torch.float32, pin_memory=True))
            except Exception:
                pin = False
            self._pin_out = pin
        return pin or None

    def search_into(self, q, k: int, scores_out, rows_out, mode: int = 0) -> None:
        """Same as ``search`` but writes into caller-provided torch tensors (host or device):
        ``scores_out`` float32 [nq,k], ``rows_out`` int64 [nq,k], both contiguous."""
        q = q.detach().float().contiguous() if not isinstance(q, np.ndarray) else np.ascontiguousarray(q, dtype=np.float32)
        nq, k = int(q.shape[0]), int(k)
        if not 0 < k <= self.ntotal:
            raise ValueError(f"top_docs={k} must satisfy 0 < k <= ntotal={self.ntotal}")
        assert tuple(scores_out.shape) == (nq, k) and tuple(rows_out.shape) == (nq, k)
        assert scores_out.is_contiguous() and rows_out.is_contiguous()
        p, keep = self._ptr(q)
        _lib.check(self._lib.kr_index_search(self._h, p, nq, k, int(scores_out.data_ptr()), int(rows_out.data_ptr()), int(mode),
                                             self._stream(q, scores_out, rows_out)))

    def search_async(self, q, k: int, scores_out, rows_out) -> None:
        """Enqueue-only half of ``search_into`` (``kr_index_search_async``): pass 1 of every 1024-query block goes onto torch's current stream and the
        call returns without waiting for the device; ``q`` (a contiguous float32 CUDA tensor), ``scores_out`` and ``rows_out`` must stay alive and
        untouched until ``finish()`` — which waits, re-answers the queries whose exactness certificate did not hold, and makes the results final."""
        import torch
        if not (torch.is_tensor(q) and q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()):
            raise ValueError("search_async needs a contiguous float32 CUDA tensor of queries")
        nq, k = int(q.shape[0]), int(k)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"expected [nq,{self.d}] queries, got {tuple(q.shape)}")
        if not 0 < k <= self.ntotal:
            raise ValueError(f"top_docs={k} must satisfy 0 < k <= ntotal={self.ntotal}")
        assert tuple(scores_out.shape) == (nq, k) and tuple(rows_out.shape) == (nq, k)
        assert scores_out.is_contiguous() and rows_out.is_contiguous() and scores_out.dtype == torch.float32 and rows_out.dtype == torch.int64
        if nq == 0:
            raise ValueError("search_async needs at least one query")              # (kr_index_search_async would return without an outstanding call)
        if not isinstance(getattr(self, "_pending", None), list):
            self._pending = []
        self._pending.append((q, scores_out, rows_out))                            # keep-alive until finish()
        _lib.check(self._lib.kr_index_search_async(self._h, int(q.data_ptr()), nq, k, int(scores_out.data_ptr()), int(rows_out.data_ptr()),
                                                   self._stream(q, scores_out, rows_out)))

    def search_coarse_async(self, q, k: int, topk_out) -> None:
        """First half of the split search of a row shard (``kr_index_search_coarse_async``): the coarse scan of at most 1024 queries, enqueue only;
        ``topk_out`` (float32 CUDA tensor [nq, k + 1]) receives this shard's k best coarse scores per query and the query's error bound — what the shards
        exchange BEFORE anybody re-ranks (``ShardedSearcher``)."""
        import torch
        if not (torch.is_tensor(q) and q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()):
            raise ValueError("search_coarse_async needs a contiguous float32 CUDA tensor of queries")
        nq, k = int(q.shape[0]), int(k)
        if q.ndim != 2 or q.shape[1] != self.d or not 0 < nq <= 1024:
            raise ValueError(f"expected [1..1024,{self.d}] queries, got {tuple(q.shape)}")
        if not 0 < k <= self.ntotal:
            raise ValueError(f"top_docs={k} must satisfy 0 < k <= ntotal={self.ntotal}")
        assert tuple(topk_out.shape) == (nq, k + 1) and topk_out.is_cuda and topk_out.is_contiguous() and topk_out.dtype == torch.float32
        if not isinstance(getattr(self, "_pending", None), list):
            self._pending = []
        self._pending.append((q, topk_out))
        _lib.check(self._lib.kr_index_search_coarse_async(self._h, int(q.data_ptr()), nq, k, int(topk_out.data_ptr()), self._stream(q)))

    def search_global_theta(self, gathered, nshards: int, theta_out) -> None:
        """``gathered``: the shards' ``topk_out`` blocks one after the other ([nshards * nq, k + 1], any rank order) -> ``theta_out`` [nq]: the bound below
        which a row of THIS shard cannot be in the global top-k (``kr_index_search_global_theta``; same stream as the two halves)."""
        assert gathered.is_cuda and gathered.is_contiguous() and theta_out.is_cuda and theta_out.is_contiguous()
        _lib.check(self._lib.kr_index_search_global_theta(self._h, int(gathered.data_ptr()), int(nshards), int(theta_out.data_ptr()), self._stream(gathered)))

    def search_rerank_async(self, theta, scores_out, rows_out) -> None:
        """Second half: certificate + exact re-rank above ``theta`` (a float32 CUDA tensor [nq], or None = the shard's own bound only); a query may come back with
        fewer than k rows, the tail then is (-inf, -1).  Final after ``finish()``."""
        assert scores_out.is_contiguous() and rows_out.is_contiguous()
        self._pending.append((theta, scores_out, rows_out))
        _lib.check(self._lib.kr_index_search_rerank_async(self._h, int(theta.data_ptr()) if theta is not None else None, int(scores_out.data_ptr()),
                                                          int(rows_out.data_ptr()), self._stream(scores_out, rows_out, *( [theta] if theta is not None else []))))

    def finish(self):
        """Finish EVERY outstanding ``search_async`` call (up to 16 may be enqueued back to back on one stream).  Returns, per call and oldest first,
        the number of its queries that pass 1 could not certify: for those the rows of the call's output tensors were re-written by passes 2 / 3 after
        anything the caller had enqueued behind the call read them (``kr_index_search_finish_ex``)."""
        flagged = (C.c_int64 * 16)()
        ncalls = C.c_int(0)
        try:
            _lib.check(self._lib.kr_index_search_finish_ex(self._h, flagged, 16, C.byref(ncalls)))
        finally:
            self._pending = []
        return [int(flagged[i]) for i in range(min(ncalls.value, 16))]

    def finish_one(self) -> int:
        """Finish the OLDEST outstanding ``search_async`` call only (``kr_index_search_finish_one``); returns the number of its queries pass 1 could not certify."""
        fl = C.c_int64(0)
        try:
            _lib.check(self._lib.kr_index_search_finish_one(self._h, C.byref(fl)))
        finally:
            if self._pending:
                self._pending.pop(0)
        return int(fl.value)

    def reconstruct_n(self, start: int, n: int) -> np.ndarray:
        out = np.empty((n, self.d), np.float32)
        _lib.check(self._lib.kr_index_get_rows(self._h, int(start), int(n), out.ctypes.data, None))
        return out

    def reconstruct_rows(self, rows) -> np.ndarray:
        """fp32 master rows of arbitrary row numbers (runs of consecutive rows are read together)."""
        rows = np.asarray(rows, np.int64).reshape(-1)
        out = np.empty((len(rows), self.d), np.float32)
        j = 0
        while j < len(rows):
            e = j + 1
            while e < len(rows) and rows[e] == rows[e - 1] + 1:
                e += 1
            out[j:e] = self.reconstruct_n(int(rows[j]), e - j)
            j = e
        return out

    # ---- stored state, exactly (native shard files) ------------------------------------------------------------------------
    @property
    def coarse_dim(self) -> int:
        return int(self._lib.kr_index_coarse_dim(self._h))

    def coarse_rows(self, start: int, n: int) -> np.ndarray:
        out = np.empty((n, self.coarse_dim), np.uint16)
        _lib.check(self._lib.kr_index_get_coarse(self._h, int(start), int(n), out.ctypes.data, None))
        return out

    def bounds(self) -> np.ndarray:
        out = np.empty(2, np.float32)
        _lib.check(self._lib.kr_index_get_bounds(self._h, out.ctypes.data))
        return out

    def add_raw(self, xf: np.ndarray, xc: np.ndarray, bounds: np.ndarray) -> None:
        xf = np.ascontiguousarray(xf, np.float32); xc = np.ascontiguousarray(xc, np.uint16); bounds = np.ascontiguousarray(bounds, np.float32)
        if xf.ndim != 2 or xf.shape[1] != self.d or xc.shape != (xf.shape[0], self.coarse_dim) or bounds.shape != (2,):
            raise ValueError("add_raw: expected xf [n,d] float32, xc [n,coarse_dim] uint16, bounds [2]")
        _lib.check(self._lib.kr_index_add_raw(self._h, xf.ctypes.data, xc.ctypes.data, int(xf.shape[0]), bounds.ctypes.data, None))

    def stats(self, reset: bool = False) -> dict:
        st = _lib.SearchStats()
        _lib.check(self._lib.kr_index_stats(self._h, C.byref(st), int(reset)))
        return {f: getattr(st, f) for f, _ in st._fields_}


def ids_to_str_rows(ext: np.ndarray) -> List[List[str]]:
    """``[[str(v) for v in row] for row in ext]`` (index.py:49 maps every hit to ``str(id)``) without 100 k Python-level ``str`` calls per
    1024-query x top-100 block: the library writes the ids as ONE ASCII buffer (``kr_format_ids``), which is decoded and split once — 7 ms instead
    of 25 ms per block on this container's host, the same lists of the same strings (tests/test_capi_and_host.py)."""
    ext = np.ascontiguousarray(ext, dtype=np.int64)
    nq, k = ext.shape
    if ext.size == 0:
        return [[] for _ in range(nq)]
    cap = 21 * ext.size
    buf = C.create_string_buffer(cap)
    written = C.c_int64(0)
    _lib.check(_lib.load().kr_format_ids(ext.ctypes.data, int(ext.size), b" ", C.addressof(buf), cap, C.byref(written)))
    flat = C.string_at(buf, written.value).decode("ascii").split(" ")
    return [flat[i:i + k] for i in range(0, len(flat), k)]


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

    def index_data(self, ids, embeddings):
        # reference order: id map first, then astype('float32'), then add (index.py:28-32)
        self._update_id_mapping(ids)
        if isinstance(embeddings, np.ndarray):
            embeddings = embeddings.astype('float32')
        self.index.add(embeddings)
        logger.info(f'Total data indexed {len(self.index_id_to_db_id)}')

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
        qd = (query_vectors if torch.is_tensor(query_vectors) else torch.from_numpy(np.ascontiguousarray(query_vectors, dtype=np.float32)))
        qd = qd.detach().to(dev, dtype=torch.float32).contiguous()       # ONE upload of all queries (nq x 4 KiB) before the first search: an upload from pageable
                                                                         # memory per block would queue behind the searches in flight on the same stream
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
            self.index.search_async(qd[s0:e0], top_docs, ps[:e0 - s0], pi[:e0 - s0])

        with torch.cuda.device(dev):
            try:
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
        return result

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

    def _update_id_mapping(self, db_ids: List):
        new_ids = np.array(db_ids, dtype=np.int64)
        self.index_id_to_db_id = np.concatenate((self.index_id_to_db_id, new_ids), axis=0)


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


# ---------------------------------------------------------------------------------------------------------
# faiss flat-index file layout.  faiss is a third-party dependency of the reference (requirements.txt:10) and
# its source is not on disk here: the layout below restates faiss 1.8 `write_index` for IndexFlat from its
# published io code (impl/index_write.cpp: fourcc, write_index_header, WRITEXBVECTOR) and is UNVERIFIED
# against a real faiss build in this environment: tests/test_capi_and_host.py pins the writer to a hand-assembled
# byte string of that field list (header 4+4+8+8+8+1+4 = 37 bytes, then the WRITEXBVECTOR count = payload bytes / 4),
# which guards the layout against regressions but is NOT a round trip through faiss.
#   u32  fourcc "IxFI"
#   i32  d ; i64 ntotal ; i64 dummy (1<<20) ; i64 dummy (1<<20) ; u8 is_trained ; i32 metric_type (0 = IP)
#   u64  number of float32 values (= ntotal * d) ; float32[ntotal * d] row-major
# ---------------------------------------------------------------------------------------------------------
_FOURCC_IXFI = struct.unpack("<I", b"IxFI")[0]


def write_faiss_flat_ip(index: FlatIPIndex, path: str) -> None:
    n, d = index.ntotal, index.d
    with open(path, "wb") as f:
        f.write(struct.pack("<I", _FOURCC_IXFI))
        f.write(struct.pack("<iqqqBi", d, n, 1 << 20, 1 << 20, 1, 0))
        f.write(struct.pack("<Q", n * d))
        for s in range(0, n, _IO_CHUNK_ROWS):
            m = min(_IO_CHUNK_ROWS, n - s)
            f.write(index.reconstruct_n(s, m).tobytes())


def read_faiss_flat_ip(path: str, device: Optional[int] = None, coarse_dtype: str = "bf16", row_range=None) -> FlatIPIndex:
    """``row_range = (rank, world)`` loads only that rank's contiguous share of the rows (``ShardedIndexer``); the returned index carries
    ``file_ntotal`` (rows in the file) and ``row_offset`` (first row held)."""
    with open(path, "rb") as f:
        (fourcc,) = struct.unpack("<I", f.read(4))
        if fourcc != _FOURCC_IXFI:
            raise ValueError(f"{path}: not a faiss IndexFlatIP file (fourcc {struct.pack('<I', fourcc)!r})")
        d, n, _, _, _trained, metric = struct.unpack("<iqqqBi", f.read(4 + 8 * 3 + 1 + 4))
        if metric != 0:
            raise ValueError(f"{path}: metric_type {metric} is not METRIC_INNER_PRODUCT")
        (nfloat,) = struct.unpack("<Q", f.read(8))
        if nfloat != n * d:
            raise ValueError(f"{path}: payload {nfloat} floats != ntotal*d = {n * d}")
        a, b = 0, n
        if row_range is not None:
            rank, world = row_range
            per = (n + world - 1) // world
            a, b = min(rank * per, n), min((rank + 1) * per, n)
            f.seek(a * d * 4, os.SEEK_CUR)
        index = FlatIPIndex(d, device=device, coarse_dtype=coarse_dtype)
        index.reserve(b - a)
        for s in range(a, b, _IO_CHUNK_ROWS):
            m = min(_IO_CHUNK_ROWS, b - s)
            buf = np.frombuffer(f.read(m * d * 4), dtype=np.float32).reshape(m, d)
            index.add(buf)
        index.file_ntotal, index.row_offset = n, a
    return index


# ---------------------------------------------------------------------------------------------------------
# Native shard files: what one rank's FlatIPIndex holds, byte for byte, so that a reload does not re-quantise.
#   bytes 0..7    magic  b"KRSHARD1"
#   i32 d ; i32 coarse_dim ; i32 coarse_dtype (0 bf16, 1 f16) ; i32 reserved = 0
#   i64 row0 (first global row) ; i64 rows ; i64 ntotal (rows of the whole corpus)
#   f32 bounds[2]   max |x - c(x)|_2 , max |c(x)|_2 over the rows of the index that wrote the file
#   f32 [rows, d]   master rows ; u16 [rows, coarse_dim]  scan copy
# ---------------------------------------------------------------------------------------------------------
SHARD_MANIFEST = "kirag_shards.json"


def _ids_crc32(ids) -> int:
    import zlib
    return int(zlib.crc32(np.ascontiguousarray(np.asarray(ids, dtype=np.int64)).tobytes()) & 0xFFFFFFFF)


def _manifest_matches(manifest_path: str, id_map) -> bool:
    """The native shards belong to the ``index_meta.faiss`` next to them: same row count and (manifests written since round 3) the same CRC-32 of
    the id map.  A directory later rewritten with the reference-format files fails this and is loaded from ``index.faiss``."""
    import json
    try:
        with open(manifest_path) as f:
            man = json.load(f)
    except Exception:
        return False
    if int(man.get("ntotal", -1)) != len(id_map):
        return False
    crc = man.get("meta_crc32")
    return crc is None or int(crc) == _ids_crc32(id_map)
_SHARD_MAGIC = b"KRSHARD1"
_SHARD_HEADER = struct.Struct("<8siiiiqqqff")


def shard_file_name(rank: int, world: int) -> str:
    return f"index_shard_{rank:04d}_of_{world:04d}.krshard"


def write_native_shard(index: FlatIPIndex, path: str, row0: int, ntotal: int) -> None:
    n, d, dc = index.ntotal, index.d, index.coarse_dim
    b = index.bounds() if n else np.zeros(2, np.float32)
    with open(path, "wb") as f:
        f.write(_SHARD_HEADER.pack(_SHARD_MAGIC, d, dc, {"bf16": 0, "f16": 1}[index.coarse_dtype], 0, int(row0), n, int(ntotal), float(b[0]), float(b[1])))
        for s in range(0, n, _IO_CHUNK_ROWS):
            f.write(index.reconstruct_n(s, min(_IO_CHUNK_ROWS, n - s)).tobytes())
        for s in range(0, n, _IO_CHUNK_ROWS):
            f.write(index.coarse_rows(s, min(_IO_CHUNK_ROWS, n - s)).tobytes())


def read_native_shards(dir_path: str, device: Optional[int] = None, coarse_dtype: str = "bf16", row_range=None) -> FlatIPIndex:
    """Rows [a, b) of the corpus (``row_range = (rank, world)``: that rank's contiguous share; None: everything) from whichever shard files hold
    them — the loading world size need not be the saving one.  The 16-bit copy is taken from the files when their dtype matches ``coarse_dtype``
    (bounds = the maximum over the files read, which is valid for any subset of their rows), otherwise the rows are re-quantised."""
    import json
    with open(os.path.join(dir_path, SHARD_MANIFEST)) as f:
        man = json.load(f)
    if man.get("format") != "krshard-1":
        raise ValueError(f"{dir_path}: unknown shard format {man.get('format')!r}")
    n, d = int(man["ntotal"]), int(man["d"])
    a, b = 0, n
    if row_range is not None:
        rank, world = row_range
        per = (n + world - 1) // world
        a, b = min(rank * per, n), min((rank + 1) * per, n)
    index = FlatIPIndex(d, device=device, coarse_dtype=coarse_dtype)
    index.reserve(b - a)
    raw = man["coarse_dtype"] == coarse_dtype
    want_code = {"bf16": 0, "f16": 1}[coarse_dtype]
    covered = a
    for sh in man["shards"]:
        r0, rows = int(sh["row0"]), int(sh["rows"])
        lo, hi = max(a, r0), min(b, r0 + rows)
        if lo >= hi:
            continue
        if lo != covered:
            raise ValueError(f"{dir_path}: rows [{covered}, {lo}) are in no shard file")
        with open(os.path.join(dir_path, sh["file"]), "rb") as f:
            magic, fd, fdc, fct, _, fr0, frows, fnt, b0, b1 = _SHARD_HEADER.unpack(f.read(_SHARD_HEADER.size))
            if magic != _SHARD_MAGIC or fd != d or fr0 != r0 or frows != rows or fnt != n:
                raise ValueError(f"{sh['file']}: header does not match the manifest")
            base_f = _SHARD_HEADER.size
            base_c = base_f + rows * d * 4
            for s0 in range(lo, hi, _IO_CHUNK_ROWS):
                m = min(_IO_CHUNK_ROWS, hi - s0)
                f.seek(base_f + (s0 - r0) * d * 4)
                xf = np.frombuffer(f.read(m * d * 4), dtype=np.float32).reshape(m, d)
                if raw and fdc == index.coarse_dim and fct == want_code:     # the FILE's own dtype code, not only the manifest's word
                    f.seek(base_c + (s0 - r0) * fdc * 2)
                    xc = np.frombuffer(f.read(m * fdc * 2), dtype=np.uint16).reshape(m, fdc)
                    index.add_raw(xf, xc, np.array([b0, b1], np.float32))
                else:
                    index.add(xf)
        covered = hi
    if covered != b:
        raise ValueError(f"{dir_path}: rows [{covered}, {b}) are in no shard file")
    index.file_ntotal, index.row_offset = n, a
    return index
