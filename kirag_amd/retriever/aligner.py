"""KiRAG-loop aligner step on the MI355X path.

The reference's ``KiRAG.filter_candidate_triples`` (``knowledge_graph/models.py:1514-1542``) re-encodes ALL candidate triples of a
question on every turn (up to 5 turns, ``models.py:1199``) in batches of 4 with a device->host copy per batch
(``retriever/retrievers.py:214-232``), then ranks them with ``torch.matmul`` + ``torch.topk`` on the host.  Here:

* ``EmbeddingCache`` — text -> embedding row, so a triple (or query) seen on an earlier turn is not encoded again.  The HIP encoder is
  deterministic and a row does not depend on the rest of its batch (tested), so cached rows are bit-identical to re-encoded ones.
* ``rank_by_similarity`` — exact top-k of ``queries @ candidates.T`` on the device through ``kr_score_topk`` (canonical scores, ties by
  candidate index), returning python lists exactly like ``models.py:1539-1542``.
* ``filter_candidate_triples`` — the reference function's body with the two calls above; ``KiRAG`` can bind it as a method unchanged
  (see INTEGRATION.md).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import _lib


class EmbeddingCache:
    """LRU map (kind, max_length, text) -> float32 embedding row.  ``max_bytes`` bounds the memory (default 1 GiB = 256k rows of 1024)."""

    def __init__(self, max_bytes: int = 1 << 30):
        self.max_bytes = int(max_bytes)
        self._rows: "OrderedDict[tuple, np.ndarray]" = OrderedDict()
        self._bytes = 0
        self.hits = 0
        self.misses = 0

    def __len__(self) -> int:
        return len(self._rows)

    def clear(self) -> None:
        self._rows.clear(); self._bytes = 0

    def embed(self, texts: Sequence[str], kind: str, max_length: Optional[int], encode: Callable[[List[str]], Tensor]) -> Tensor:
        """Embeddings of ``texts`` in order; ``encode(list_of_missing_texts) -> Tensor[m, H]`` is called once for the cache misses
        (duplicates inside ``texts`` are encoded once)."""
        keys = [(kind, max_length, t) for t in texts]
        missing: "OrderedDict[tuple, None]" = OrderedDict()
        for key in keys:
            if key in self._rows:
                self._rows.move_to_end(key); self.hits += 1
            elif key not in missing:
                missing[key] = None; self.misses += 1
        if missing:
            new = encode([key[2] for key in missing]).detach().float().cpu().numpy()
            for key, row in zip(missing, new):
                row = np.array(row, dtype=np.float32, copy=True)
                self._rows[key] = row; self._bytes += row.nbytes
        out = torch.from_numpy(np.stack([self._rows[key] for key in keys], axis=0))
        while self._bytes > self.max_bytes and len(self._rows) > len(set(keys)):
            _, old = self._rows.popitem(last=False); self._bytes -= old.nbytes
        return out


def rank_by_similarity(queries_embeddings: Union[Tensor, np.ndarray], candidates_embeddings: Union[Tensor, np.ndarray], k: int,
                       device: Optional[int] = None) -> Tuple[List[List[int]], List[List[float]]]:
    """``torch.topk(queries @ candidates.T, k=min(k, n), dim=1)`` (``models.py:1532-1542``) as (indices, scores) python lists, computed on
    the device: exact canonical scores, descending, ties by candidate index."""
    def prep(a):
        if isinstance(a, np.ndarray):
            return np.ascontiguousarray(a, dtype=np.float32), None
        a = a.detach().float().contiguous()
        return a, (a.device.index if a.is_cuda else None)
    q, dq = prep(queries_embeddings)
    x, dx = prep(candidates_embeddings)
    if q.ndim != 2 or x.ndim != 2 or q.shape[1] != x.shape[1]:
        raise ValueError(f"expected [nq,d] and [n,d], got {tuple(q.shape)} and {tuple(x.shape)}")
    nq, d = int(q.shape[0]), int(q.shape[1])
    n = int(x.shape[0])
    k = min(int(k), n)
    if device is None:
        device = dq if dq is not None else (dx if dx is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0))
    scores = np.empty((nq, k), np.float32); rows = np.empty((nq, k), np.int64)
    ptr = lambda a: a.ctypes.data if isinstance(a, np.ndarray) else int(a.data_ptr())
    _lib.check(_lib.load().kr_score_topk(ptr(q), nq, ptr(x), n, d, k, scores.ctypes.data, rows.ctypes.data, int(device), None))
    return rows.tolist(), scores.tolist()


def filter_candidate_triples(aligner, question: str, reasoning_chains_texts: List[List[str]], triples_texts: List[str],
                             num_candidate_triples: int, cache: Optional[EmbeddingCache] = None,
                             query_max_length: int = 256, triple_max_length: int = 128):
    """Body of ``KiRAG.filter_candidate_triples`` (``models.py:1514-1542``) after its text preparation:
    ``aligner`` is the ``DenseRetriever`` (``self.aligner``), ``reasoning_chains_texts = self.get_reasoning_chains_texts(chains)``,
    ``triples_texts = [self.get_triple_text(t) for t in triples]``.  Returns ``(topk_indices, topk_scores)`` as lists of lists."""
    queries = ["{}\nknowledge triples: {}.".format(question, ". ".join(texts)) for texts in reasoning_chains_texts]   # models.py:1526
    if cache is None:
        queries_embeddings = aligner.calculate_query_embeddings(queries=queries, max_length=query_max_length)
        triples_embeddings = aligner.calculate_document_embeddings(documents=triples_texts, max_length=triple_max_length)
    else:
        queries_embeddings = cache.embed(queries, "query", query_max_length,
                                         lambda t: aligner.calculate_query_embeddings(queries=t, max_length=query_max_length))
        triples_embeddings = cache.embed(triples_texts, "doc", triple_max_length,
                                         lambda t: aligner.calculate_document_embeddings(documents=t, max_length=triple_max_length))
    return rank_by_similarity(queries_embeddings, triples_embeddings, min(num_candidate_triples, len(triples_texts)))
