"""Search half of the oracle (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates ``retriever/index.py`` (``Indexer.index_data`` :26-34, ``search_knn`` :36-53,
``_update_id_mapping`` :81-83) on top of an exact inner-product top-k.  The arithmetic of the
reference lives in faiss-cpu==1.8.0.post1 ``IndexFlatIP`` (not vendored, not installable here):
**parity with faiss is unpinned**; semantics restated from its published behaviour.

Scorers:
  * ``search_canonical``  — ctypes call into ``search_c.c``: the canonical score = the EXACT inner product of
    the fp32 inputs rounded once to fp32 (round-to-nearest-even) — a definition that does not depend on any
    summation order — and the tie rule (score desc, internal row asc).  This is what the HIP path must match
    bit for bit.  ``dot_exact`` (super-accumulator) and ``dot_fraction`` (Python rationals) are the two
    independent formulations the C fast path is pinned against.
  * ``search_f64``        — numpy fp64 GEMM, rounded to fp32, same tie rule.  Equal to the canonical
    result except when an fp64 sum lands within ~1e-13 relative of an fp32 rounding boundary; used
    for sizes where the scalar C loop is too slow.
  * ``search_sgemm``      — fp32 BLAS sgemm + argpartition in 1024-query blocks: the CPU stand-in for
    ``faiss.IndexFlatIP.search`` used as ``bench.py``'s ``cpu_baseline`` (kind "port").
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "search_c.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        for nm in ("kr_oracle_dot", "kr_oracle_dot_exact", "kr_oracle_dot_f32"):
            getattr(L, nm).restype = ctypes.c_float
            getattr(L, nm).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.kr_oracle_scores_all.restype = None
        L.kr_oracle_scores_all.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.kr_oracle_search.restype = ctypes.c_int
        L.kr_oracle_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.kr_oracle_scores_at.restype = None
        L.kr_oracle_scores_at.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.kr_oracle_f32_to_bf16.restype = ctypes.c_uint16
        L.kr_oracle_f32_to_bf16.argtypes = [ctypes.c_float]
        _LIB = L
    return _LIB


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def search_canonical(q: np.ndarray, x: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    q = _c(q, np.float32); x = _c(x, np.float32)
    nq, d = q.shape
    n = x.shape[0]
    s = np.empty((nq, k), np.float32); i = np.empty((nq, k), np.int64)
    rc = lib().kr_oracle_search(q.ctypes.data, nq, x.ctypes.data, n, d, k, s.ctypes.data, i.ctypes.data)
    if rc == -2:
        raise ValueError(f"fewer than k={k} rows have a real (non-NaN) score")
    if rc != 0:
        raise ValueError(f"k={k} must satisfy 0 < k <= ntotal={n}")
    return s, i


def dot(q: np.ndarray, x: np.ndarray, which: str = "canonical") -> float:
    """One inner product by ``which`` in {"canonical" (sequential fp64 + certified rounding), "exact" (super-accumulator), "f32" (fp32 loop)}."""
    q = _c(q, np.float32); x = _c(x, np.float32)
    fn = {"canonical": lib().kr_oracle_dot, "exact": lib().kr_oracle_dot_exact, "f32": lib().kr_oracle_dot_f32}[which]
    return float(np.float32(fn(q.ctypes.data, x.ctypes.data, int(q.shape[0]))))


def scores_all(q: np.ndarray, x: np.ndarray, which: str = "canonical") -> np.ndarray:
    """All-pairs scores [nq, n] by one of the three C formulations (see ``dot``)."""
    q = _c(q, np.float32); x = _c(x, np.float32)
    out = np.empty((q.shape[0], x.shape[0]), np.float32)
    lib().kr_oracle_scores_all(q.ctypes.data, q.shape[0], x.ctypes.data, x.shape[0], q.shape[1],
                               {"canonical": 0, "exact": 1, "f32": 2}[which], out.ctypes.data)
    return out


def dot_fraction(q: np.ndarray, x: np.ndarray) -> np.float32:
    """Third, fully independent formulation: the inner product in Python rationals (exact), then ONE round-to-nearest-even
    to fp32 done on integers.  Slow (pure Python): small cases and the golden vectors only."""
    from fractions import Fraction
    tot = Fraction(0)
    for a, b in zip(np.asarray(q, np.float32).tolist(), np.asarray(x, np.float32).tolist()):
        tot += Fraction(a) * Fraction(b)                 # Fraction(float) is exact
    return round_fraction_to_f32(tot)


def round_fraction_to_f32(v) -> np.float32:
    """RN-even of a rational to fp32 (normal, subnormal, overflow to inf, signed zero for tiny non-zero values)."""
    from fractions import Fraction
    if v == 0:
        return np.float32(0.0)
    neg = v < 0
    a = -v if neg else v
    # exponent e with 2^e <= a < 2^(e+1)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a:
        e += 1
    qexp = max(e - 23, -149)                             # exponent of the fp32 quantum
    scaled = a / (Fraction(2) ** qexp)
    m = scaled.numerator // scaled.denominator
    rem = scaled - m
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (m & 1)):
        m += 1
    val = np.float64(m) * np.float64(2.0) ** qexp if qexp > -1000 else 0.0     # m <= 2^24: exact in fp64
    with np.errstate(over="ignore"):
        f = np.float32(val)                              # exact (m * 2^qexp is representable) or inf on overflow
    return np.float32(-f) if neg else f


def scores_at(q: np.ndarray, x: np.ndarray, rows: np.ndarray) -> np.ndarray:
    q = _c(q, np.float32); x = _c(x, np.float32); rows = _c(rows, np.int64)
    out = np.empty(rows.shape, np.float32)
    lib().kr_oracle_scores_at(q.ctypes.data, q.shape[0], x.ctypes.data, q.shape[1], rows.ctypes.data,
                              rows.shape[1], out.ctypes.data)
    return out


def topk_desc(scores: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per row: k best by (score desc, column asc).  scores float32 [nq, n]."""
    nq, n = scores.shape
    if not 0 < k <= n:
        raise ValueError(f"k={k} must satisfy 0 < k <= ntotal={n}")
    out_s = np.empty((nq, k), np.float32); out_i = np.empty((nq, k), np.int64)
    cols = np.arange(n)
    for r in range(nq):
        row = scores[r]
        if k < n:
            kth = np.partition(row, n - k)[n - k]
            cand = np.nonzero(row >= kth)[0]
        else:
            cand = cols
        order = np.lexsort((cand, -row[cand].astype(np.float64)))[:k]
        out_i[r] = cand[order]; out_s[r] = row[cand[order]]
    return out_s, out_i


def search_f64(q: np.ndarray, x: np.ndarray, k: int, block: int = 256) -> Tuple[np.ndarray, np.ndarray]:
    q64 = np.asarray(q, np.float32).astype(np.float64)
    x64t = np.asarray(x, np.float32).astype(np.float64).T
    ss, ii = [], []
    for b in range(0, len(q64), block):
        sc = (q64[b:b + block] @ x64t).astype(np.float32)
        s, i = topk_desc(sc, k)
        ss.append(s); ii.append(i)
    return np.concatenate(ss), np.concatenate(ii)


def search_sgemm(q: np.ndarray, x: np.ndarray, k: int, index_batch_size: int = 1024):
    """fp32 sgemm + argpartition, 1024-query blocks as retriever/index.py:39-47 (faiss stand-in)."""
    q = np.asarray(q, np.float32); x = np.asarray(x, np.float32)
    n = x.shape[0]
    ss, ii = [], []
    for b in range(0, len(q), index_batch_size):
        sc = q[b:b + index_batch_size] @ x.T
        if k < n:
            part = np.argpartition(-sc, k - 1, axis=1)[:, :k]
        else:
            part = np.broadcast_to(np.arange(n), sc.shape).copy()
        ps = np.take_along_axis(sc, part, axis=1)
        order = np.argsort(-ps, axis=1, kind="stable")
        ii.append(np.take_along_axis(part, order, axis=1)); ss.append(np.take_along_axis(ps, order, axis=1))
    return np.concatenate(ss), np.concatenate(ii)


def merge_shards(scores: Sequence[np.ndarray], ids: Sequence[np.ndarray], k: int):
    """Merge per-shard (score desc) lists: k best by (score desc, id asc).  ids are GLOBAL row numbers."""
    s = np.concatenate(scores, axis=1); i = np.concatenate(ids, axis=1)
    out_s = np.empty((s.shape[0], k), np.float32); out_i = np.empty((s.shape[0], k), np.int64)
    for r in range(s.shape[0]):
        order = np.lexsort((i[r], -s[r].astype(np.float64)))[:k]
        out_s[r] = s[r][order]; out_i[r] = i[r][order]
    return out_s, out_i


class OracleIndexer:
    """retriever/index.py:17-83 restated over ``search_canonical`` (inner_product, flat only)."""

    def __init__(self, vector_sz: int, metric: str = "inner_product", n_subquantizers: int = 0, n_bits: int = 8):
        if metric != "inner_product" or n_subquantizers > 0:
            raise NotImplementedError("oracle covers the IndexFlatIP path only (the only one the reference's callers use)")
        self.d = vector_sz
        self.x = np.empty((0, vector_sz), np.float32)
        self.index_id_to_db_id = np.empty((0), dtype=np.int64)

    def index_data(self, ids, embeddings):
        self.index_id_to_db_id = np.concatenate((self.index_id_to_db_id, np.array(ids, dtype=np.int64)), axis=0)
        self.x = np.concatenate([self.x, np.asarray(embeddings).astype("float32")], axis=0)

    def search_knn(self, query_vectors, top_docs: int, index_batch_size: int = 1024, verbose: bool = True, exact_c: bool = True):
        qv = np.asarray(query_vectors).astype("float32")
        result: List[Tuple[List[str], np.ndarray]] = []
        fn = search_canonical if exact_c else search_f64
        for b in range(0, len(qv), index_batch_size):
            s, i = fn(qv[b:b + index_batch_size], self.x, top_docs)
            for r in range(len(s)):
                result.append(([str(self.index_id_to_db_id[j]) for j in i[r]], s[r]))
        return result


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bf16 bit pattern (uint16); vectorised twin of kr_oracle_f32_to_bf16."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)
