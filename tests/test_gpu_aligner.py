"""GPU parity of the KiRAG-loop aligner step (kirag_amd/retriever/aligner.py, kr_score_topk) against the oracle:
indices identical and scores bit-identical to oracle.search_np.search_canonical; agreement with the reference's own
torch.matmul + torch.topk formulation (knowledge_graph/models.py:1532-1538) up to fp32 rounding."""
import numpy as np
import pytest
import torch

from oracle import search_np as S

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("nq,n,d,k", [(1, 20, 1024, 20), (2, 700, 1024, 20), (2, 37, 128, 20), (5, 3000, 256, 64), (1, 1, 64, 1), (3, 1025, 64, 1024)])
def test_rank_by_similarity_matches_canonical_oracle(nq, n, d, k):
    from kirag_amd.retriever.aligner import rank_by_similarity
    rng = np.random.default_rng(nq * 1000 + n)
    q, x = _unit(rng, nq, d), _unit(rng, n, d)
    idx, sc = rank_by_similarity(torch.from_numpy(q), torch.from_numpy(x), k)
    kk = min(k, n)
    so, io = S.search_canonical(q, x, kk)
    assert np.array_equal(np.asarray(idx, np.int64), io)
    assert np.array_equal(np.asarray(sc, np.float32).view(np.uint32), so.view(np.uint32))
    # the reference formulation (models.py:1532-1538): same sets, scores within fp32 rounding
    ref_s, ref_i = torch.topk(torch.from_numpy(q) @ torch.from_numpy(x).T, k=kk, dim=1)
    assert np.abs(np.asarray(sc, np.float32) - ref_s.numpy()).max() < 2e-6
    for a, b in zip(idx, ref_i.tolist()):
        assert len(set(a) ^ set(b)) <= 2                    # only near-ties at the cut may differ
    # device-resident inputs take the same path
    idx2, sc2 = rank_by_similarity(torch.from_numpy(q).cuda(), torch.from_numpy(x).cuda(), k)
    assert idx2 == idx and sc2 == sc


def test_duplicate_candidates_tie_rule():
    from kirag_amd.retriever.aligner import rank_by_similarity
    rng = np.random.default_rng(5)
    x = _unit(rng, 10, 64)
    x[7] = x[2]; x[4] = x[2]
    idx, sc = rank_by_similarity(x[2:3], x, 4)
    assert idx[0][:3] == [2, 4, 7] and sc[0][0] == sc[0][1] == sc[0][2]     # equal scores: lower candidate index first


def test_filter_candidate_triples_with_cache_equals_without(golden):
    """End to end through the HIP encoder: cached turns return exactly what a fresh encode returns, and encode only the new triples."""
    from types import SimpleNamespace
    from oracle import encoder_np as E
    from kirag_amd.retriever.aligner import EmbeddingCache, filter_candidate_triples
    from kirag_amd.retriever.encoders import HipBertForward
    cfg = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                          max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    enc = HipBertForward(cfg, 0); enc.load_state(E.synth_weights(128, 2, 512, 1000, 512, seed=5))

    class Aligner:                      # DenseRetriever-shaped: texts -> token ids by a toy hash tokenizer, then the HIP encoder
        calls = []

        def _embed(self, texts, max_length):
            self.calls.append(len(texts))
            L = max(len(t.split()) for t in texts) + 2
            ids = np.zeros((len(texts), L), np.int64); mask = np.zeros_like(ids)
            for i, t in enumerate(texts):
                toks = [101] + [5 + (sum(map(ord, w)) % 990) for w in t.split()][: max_length - 2] + [102]
                ids[i, : len(toks)] = toks; mask[i, : len(toks)] = 1
            return torch.from_numpy(enc.forward_np(ids, mask, 0))

        def calculate_query_embeddings(self, queries, max_length=None, **kw):
            return self._embed(queries, max_length)

        def calculate_document_embeddings(self, documents, max_length=None, **kw):
            return self._embed(documents, max_length)

    al = Aligner()
    triples = [f"<entity {i}; relation {i % 7}; object {i * 3}>" for i in range(50)]
    chains = [["<entity 1; relation 1; object 3>"], ["<entity 2; relation 2; object 6>", "<entity 9; relation 2; object 27>"]]
    fresh = filter_candidate_triples(al, "which object ?", chains, triples, 20)
    cache = EmbeddingCache()
    al.calls.clear()
    turn1 = filter_candidate_triples(al, "which object ?", chains, triples, 20, cache=cache)
    assert turn1 == fresh and al.calls == [2, 50]
    al.calls.clear()
    more = triples + [f"<new {i}; r; o>" for i in range(5)]
    turn2 = filter_candidate_triples(al, "which object ?", chains, more, 20, cache=cache)
    assert al.calls == [5]                                   # queries and the 50 old triples come from the cache
    assert turn2 == filter_candidate_triples(al, "which object ?", chains, more, 20)
