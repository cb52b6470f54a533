"""CPU-side checks of the KiRAG-loop aligner helpers (kirag_amd/retriever/aligner.py): the embedding cache and the text contract of
filter_candidate_triples (reference knowledge_graph/models.py:1514-1542).  The ranking itself needs the GPU (tests/test_gpu_aligner.py)."""
import numpy as np
import pytest
import torch

from kirag_amd import _lib
from kirag_amd.retriever.aligner import EmbeddingCache, filter_candidate_triples


def _fake_encode(calls):
    def encode(texts):
        calls.append(list(texts))
        return torch.tensor([[float(len(t)), float(sum(map(ord, t)) % 97), 1.0, 0.0] for t in texts])
    return encode


def test_embedding_cache_encodes_each_text_once_and_keeps_order():
    calls = []
    c = EmbeddingCache()
    a = c.embed(["bb", "a", "bb", "ccc"], "doc", 128, _fake_encode(calls))
    assert calls == [["bb", "a", "ccc"]]                      # duplicates inside one request are encoded once
    assert a.shape == (4, 4) and torch.equal(a[0], a[2]) and a[1, 0] == 1.0 and a[3, 0] == 3.0
    b = c.embed(["ccc", "dddd", "a"], "doc", 128, _fake_encode(calls))
    assert calls[-1] == ["dddd"]                              # only the miss
    assert torch.equal(b[0], a[3]) and torch.equal(b[2], a[1])
    c.embed(["a"], "doc", 64, _fake_encode(calls))            # another max_length is another key (truncation differs)
    c.embed(["a"], "query", 128, _fake_encode(calls))         # and so is the other prefix kind
    assert calls[-2:] == [["a"], ["a"]]
    assert c.hits == 2 and c.misses == 6 and len(c) == 6


def test_embedding_cache_lru_bound():
    calls = []
    c = EmbeddingCache(max_bytes=3 * 16)                      # three float32[4] rows
    for t in ["t1", "t2", "t3", "t4"]:
        c.embed([t], "doc", 8, _fake_encode(calls))
    assert len(c) == 3
    c.embed(["t1"], "doc", 8, _fake_encode(calls))            # evicted -> encoded again
    assert calls[-1] == ["t1"]
    c.embed(["t4"], "doc", 8, _fake_encode(calls))            # still resident
    assert calls[-1] == ["t1"]


def test_filter_candidate_triples_builds_the_reference_query_strings(monkeypatch):
    seen = {}

    class FakeAligner:
        def calculate_query_embeddings(self, queries, max_length=None, **kw):
            seen["queries"] = list(queries); seen["qlen"] = max_length
            return torch.eye(4)[: len(queries)]

        def calculate_document_embeddings(self, documents, max_length=None, **kw):
            seen["docs"] = list(documents); seen["dlen"] = max_length
            return torch.eye(4)[: len(documents)]

    import kirag_amd.retriever.aligner as A
    monkeypatch.setattr(A, "rank_by_similarity", lambda q, x, k, device=None: ("idx", (tuple(q.shape), tuple(x.shape), k)))
    out = filter_candidate_triples(FakeAligner(), "who?", [["<a; r; b>", "<b; r; c>"], []], ["<x; y; z>", "<p; q; r>", "<s; t; u>"], 20)
    # models.py:1526  "{}\nknowledge triples: {}.".format(question, ". ".join(texts))
    assert seen["queries"] == ["who?\nknowledge triples: <a; r; b>. <b; r; c>.", "who?\nknowledge triples: ."]
    assert seen["qlen"] == 256 and seen["dlen"] == 128 and seen["docs"] == ["<x; y; z>", "<p; q; r>", "<s; t; u>"]
    assert out == ("idx", ((2, 4), (3, 4), 3))                # k = min(num_candidate_triples, num_triples)  (models.py:1536)


def test_score_topk_argument_validation_needs_no_device():
    lib = _lib.load()
    q = np.zeros((2, 8), np.float32); x = np.zeros((5, 8), np.float32)
    s = np.zeros((2, 3), np.float32); r = np.zeros((2, 3), np.int64)
    assert lib.kr_score_topk(q.ctypes.data, 2, x.ctypes.data, 5, 6, 3, s.ctypes.data, r.ctypes.data, 0, None) == -22    # d % 4 != 0
    assert lib.kr_score_topk(q.ctypes.data, 2, x.ctypes.data, 5, 8, 6, s.ctypes.data, r.ctypes.data, 0, None) == -22    # k > n
    assert lib.kr_score_topk(q.ctypes.data, 0, x.ctypes.data, 5, 8, 3, s.ctypes.data, r.ctypes.data, 0, None) == -22    # nq = 0
    assert lib.kr_score_topk(None, 2, x.ctypes.data, 5, 8, 3, s.ctypes.data, r.ctypes.data, 0, None) == -22


def test_prefetch_map_order_bound_and_errors():
    import threading
    import time
    from kirag_amd.utils import prefetch_map
    started = []

    def fn(i):
        started.append(i)
        time.sleep(0.01)
        return i * i

    it = prefetch_map(fn, range(10), depth=2)
    first = next(it)
    time.sleep(0.1)
    assert first == 0 and len(started) <= 4          # the worker runs at most depth (+1 in flight) ahead of the consumer
    assert [first] + list(it) == [i * i for i in range(10)]

    def bad(i):
        if i == 3:
            raise ValueError("boom")
        return i

    got = []
    with pytest.raises(ValueError, match="boom"):
        for x in prefetch_map(bad, range(6)):
            got.append(x)
    assert got == [0, 1, 2]
    n0 = threading.active_count()
    g = prefetch_map(fn, range(1000), depth=1); next(g); g.close()      # abandoning the generator stops the worker
    time.sleep(0.5)
    assert threading.active_count() <= n0 + 1


def test_pool_map_tokenizer_worker_processes_keep_order_and_match_in_process_collate(tmp_path):
    """SURVEY 8f-4 multi-process tokenisation: batches tokenised by `python -m kirag_amd.tokenize_worker` child processes come back in order and
    equal the in-process collator output (dataset/collators.py:59-81,143-145 semantics); a worker-side failure surfaces at the consumer."""
    import numpy as np
    import pytest
    import torch
    from kirag_amd.bench_support import wordpiece_tokenizer
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd.collators import E5Collator
    with open(tmp_path / "vocab.txt", "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "passage", ":", "hello", "world", "foo", "bar", "##s"]) + "\n")
    tok = wordpiece_tokenizer(str(tmp_path / "vocab.txt"))
    col = E5Collator(tokenizer=tok, query_maxlength=16, doc_maxlength=12)
    texts = [" ".join(np.random.default_rng(i).choice(["hello", "world", "foo", "bars", "bar"], 1 + i % 14)) for i in range(103)]

    def mk(s):
        return texts[s:s + 10], list(range(s, min(s + 10, 103)))
    out = list(CC.pool_map(mk, col, range(0, 103, 10), 3, 2))
    assert len(out) == 11
    for (enc, ids), s in zip(out, range(0, 103, 10)):
        ref = col.encode_doc(texts[s:s + 10])
        assert torch.equal(enc["input_ids"], ref["input_ids"]) and torch.equal(enc["attention_mask"], ref["attention_mask"])
        assert enc["input_ids"].dtype == torch.int64 and ids == list(range(s, min(s + 10, 103)))
    with pytest.raises(RuntimeError) as ei:
        list(CC.pool_map(lambda s: ([], [s]), col, range(3), 2, 1))          # an empty batch: the collator raises inside the worker
    assert "text_list is None or an empty" in str(ei.value)
