"""ctypes face of ``kr_index`` (include/kirag_amd.h): ``FlatIPIndex`` is the object behind ``Indexer.index`` — the subset of ``faiss.IndexFlatIP`` the reference
touches (``retriever/index.py:13-14,31-32,47``: ``ntotal``, ``d``, ``is_trained``, ``add``, ``search``) living in MI355X HBM — plus the enqueue-only / split forms
the pipelined and row-sharded searches use, and ``ids_to_str_rows`` (the bulk form of ``index.py:49``'s ``str(id)`` per hit)."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

from .. import _lib


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

    def prepare(self, nq: int = 1, k: int = 10) -> None:
        """``kr_index_prepare``: allocate the workspaces of an nq-query top-k search and (for small blocks on a large index) build / extend the int8 copy NOW,
        so that the first such search costs what every later one costs.  No-op on an empty index."""
        k = max(1, min(int(k), self.ntotal)) if self.ntotal else int(k)
        stream = None
        try:
            import torch
            if torch.cuda.is_available():
                stream = int(torch.cuda.current_stream(self.device).cuda_stream)     # behind the adds the caller queued on its current stream
        except ImportError:
            pass
        _lib.check(self._lib.kr_index_prepare(self._h, int(nq), k, stream))
        self._small_out(1, 1)          # the pinned result / query scratch of small calls (three pinned allocations: 0.25 ms of a first 0.2-ms search otherwise)

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
        q = self._local(q)
        p, keep = self._ptr(q)
        self._pending = []          # kr_index_search drains the handle's outstanding calls (finish_search): their keep-alives go with them
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

    def _local(self, q):
        """queries as this index's device can read them in stream order: a CUDA tensor on another GPU is moved by torch (which orders the copy on both devices)"""
        if not isinstance(q, np.ndarray) and getattr(q, "is_cuda", False) and q.device.index != self.device:
            import torch
            return q.to(torch.device("cuda", self.device))
        return q

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
        q = self._local(q)
        p, keep = self._ptr(q)
        self._pending = []
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
        if self._pending:            # ONE keep-alive entry per C-side call: the rerank half belongs to the coarse half's entry (finish_one pops one per call)
            self._pending[-1] = self._pending[-1] + (theta, scores_out, rows_out)
        else:
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


try:                                   # optional C extension (kirag_amd/csrc/fastids.c, built by the same Makefile when Python.h is present)
    from .. import _fastids
except ImportError:                    # not built / another interpreter: the kr_format_ids + split form below gives the same lists
    _fastids = None


def ids_to_str_rows(ext: np.ndarray) -> List[List[str]]:
    """``[[str(v) for v in row] for row in ext]`` (index.py:49 maps every hit to ``str(id)``) without 100 k Python-level ``str`` calls per
    1024-query x top-100 block.  With the ``_fastids`` extension the nested lists are built in one C loop (PyUnicode_New + memcpy per id: ~4 ms per block on
    this container's host, where the Python comprehension takes 18 ms); without it the library writes the ids as ONE ASCII buffer (``kr_format_ids``),
    which is decoded and split once (15 ms).  The same lists of the same strings either way (tests/test_capi_and_host.py)."""
    ext = np.ascontiguousarray(ext, dtype=np.int64)
    nq, k = ext.shape
    if ext.size == 0:
        return [[] for _ in range(nq)]
    if _fastids is not None:
        return _fastids.ids_to_str_rows(memoryview(ext).cast("B"), nq, k)
    return _ids_to_str_rows_ascii(ext)


def _ids_to_str_rows_ascii(ext: np.ndarray) -> List[List[str]]:
    nq, k = ext.shape
    cap = 21 * ext.size
    buf = C.create_string_buffer(cap)
    written = C.c_int64(0)
    _lib.check(_lib.load().kr_format_ids(ext.ctypes.data, int(ext.size), b" ", C.addressof(buf), cap, C.byref(written)))
    flat = C.string_at(buf, written.value).decode("ascii").split(" ")
    return [flat[i:i + k] for i in range(0, len(flat), k)]
