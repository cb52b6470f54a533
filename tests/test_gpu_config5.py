"""BASELINE config 5 AT SIZE — the KiRAG iterative loop's retrieval turn (reference: ``knowledge_graph/models.py:1514-1542`` aligner step,
``:1631-1676`` retrieve step) on one GPU:

  * a resident 5M x 1024 index (Gaussian and e5like corpora), 1-2 query vectors per hop, top-10 — the HBM-bound <= 32-query stream kernel
    (``k_coarse_q32``) at the row count where round 2 found a size-dependent schedule bug that no 40k-row test could see;
  * the aligner step with the FULL e5-large-shape encoder: 1-2 chain queries of up to 256 tokens + up to 512 candidate triples (max_length 128,
    ``models.py:1528-1531``) through ``DenseRetriever`` -> ``filter_candidate_triples`` -> ``kr_score_topk``, three hops with the
    ``EmbeddingCache`` (later hops encode only the new triples).

Checkers: every hop's retrieval ids / scores against the kernel-independent fp32 sgemm + top-k over the REGENERATED corpus (``tests/indep_check.py``)
plus canonical score bits of the returned rows (C oracle); the triple ranking against the oracle's canonical top-k bit for bit; a cached turn against a
fresh turn bit for bit; the HIP encoder against the same module's PyTorch fp32 forward (rocBLAS) within the encoder tolerances."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from oracle import search_np as S

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import indep_check as IC  # noqa: E402

pytestmark = pytest.mark.gpu

N_ROWS, D = 5_000_000, 1024


def _vocab_and_tokenizer(td):
    from kirag_amd.bench_support import wordpiece_tokenizer
    rng = np.random.default_rng(0)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))

    def rand_words(k, lo, hi):
        out = set()
        while len(out) < k:
            out.add("".join(rng.choice(letters, int(rng.integers(lo, hi)))))
        return sorted(out)
    words = rand_words(20000, 3, 9); pieces = ["##" + w for w in rand_words(10517, 2, 5)]
    with open(os.path.join(td, "vocab.txt"), "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words + pieces) + "\n")        # 30522 entries
    return wordpiece_tokenizer(os.path.join(td, "vocab.txt")), words


def _full_size_aligner(tok):
    """DenseRetriever over the full-size E5Encoder (eval = HIP path, train = the inherited PyTorch fp32 path: the independent encoder reference)."""
    import torch.nn as nn
    from transformers import BertConfig
    from kirag_amd.collators import E5Collator
    from kirag_amd.retriever.encoders import E5Encoder
    from kirag_amd.retriever.retrievers import BaseRetriever, DenseRetriever
    cfg = BertConfig(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                     max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    enc = E5Encoder(cfg, add_pooling_layer=False)
    with torch.no_grad():                       # HF's init leaves biases 0 and LayerNorm at (1, 0): make every parameter matter
        g = torch.Generator().manual_seed(1)
        for n, p in enc.named_parameters():
            if n.endswith("LayerNorm.weight"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias"):
                p.add_(0.02 * torch.randn(p.shape, generator=g))
    enc = enc.cuda().eval()

    class Ret(BaseRetriever):
        def __init__(self, encoder):
            nn.Module.__init__(self)
            self.encoder = encoder
            self.norm_query = self.norm_doc = False
            self.temperature, self.local_rank, self.world_size = 1.0, -1, 1
    col = E5Collator(tokenizer=tok, query_maxlength=256, doc_maxlength=128)
    return DenseRetriever(retriever=Ret(enc), collator=col, indexer=None, corpus=None, batch_size=4), enc, col


def _torch_fp32(enc, col, texts, which, max_length):
    """the same module's PyTorch forward (train mode: BertModel.forward -> average_pool -> normalize, encoders.py:67-77), fp32 on the device"""
    a = (col.encode_query if which == "query" else col.encode_doc)(texts, max_length=max_length)
    enc.train()
    try:
        with torch.no_grad():
            outs = [enc(a["input_ids"][s:s + 64].cuda(), a["attention_mask"][s:s + 64].cuda()) for s in range(0, len(texts), 64)]
    finally:
        enc.eval()
    return torch.cat(outs).cpu().numpy()


@pytest.mark.parametrize("kind", ["gaussian", "e5like"])
def test_config5_three_hops_at_size(kind):
    from kirag_amd.bench_support import CorpusDist
    from kirag_amd.retriever.aligner import EmbeddingCache, filter_candidate_triples
    from kirag_amd.retriever.index import FlatIPIndex
    dev = torch.device("cuda:0")
    # ---- the resident 5M-row index ------------------------------------------------------------------------------------------------------------
    cd = CorpusDist(kind, D, dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    ix = FlatIPIndex(D, device=0); ix.reserve(N_ROWS)
    head = None
    for s0 in range(0, N_ROWS, 250_000):
        x = cd.rows(min(250_000, N_ROWS - s0), g); ix.add(x)
        if head is None:
            head = x[:64].clone()
        del x
    gq = torch.Generator(device=dev); gq.manual_seed(2)
    planted = cd.queries_near(head[:6], gq)                                        # rows 0..5 are the known nearest neighbours
    with tempfile.TemporaryDirectory() as td:
        tok, words = _vocab_and_tokenizer(td)
        aligner, enc, col = _full_size_aligner(tok)
        rng = np.random.default_rng(5)
        wa = np.array(words)

        def triple():
            k = int(rng.integers(4, 40))                                            # 6 .. ~60 tokens; a few long ones are truncated at 128
            w = rng.choice(wa, k)
            return "<" + " ".join(w[: k // 3]) + "; " + " ".join(w[k // 3: 2 * k // 3]) + "; " + " ".join(w[2 * k // 3:]) + ">"
        all_triples = [triple() for _ in range(512)]
        all_triples[7] = "<" + " ".join(rng.choice(wa, 200)) + "; r; o>"             # longer than max_length = 128: truncated
        question = "which " + " ".join(rng.choice(wa, 12)) + " ?"
        cache = EmbeddingCache()
        searched_q, searched = [], []
        chains = [[]]
        enc_calls = []
        orig = aligner._embed

        def counting(texts, which, max_length, verbose, **kw):
            enc_calls.append((which, len(texts)))
            return orig(texts, which, max_length, verbose, **kw)
        aligner._embed = counting
        for hop in range(3):
            n_tr = (400, 456, 512)[hop]
            triples = all_triples[:n_tr]
            chain_texts = [c if c else ["none"] for c in chains]                    # 1 chain at hop 0, 2 afterwards
            # ---- aligner step: cached turn == fresh turn, ranking == oracle ---------------------------------------------------------------------
            enc_calls.clear()
            got = filter_candidate_triples(aligner, question, chain_texts, triples, 20, cache=cache)
            calls_cached = list(enc_calls)
            fresh = filter_candidate_triples(aligner, question, chain_texts, triples, 20)
            assert got == fresh, f"hop {hop}: cached turn differs from a fresh turn"
            new_tr = n_tr - (0, 400, 456)[hop]
            assert ("doc", new_tr) in calls_cached and all(n <= max(new_tr, 2) for _, n in calls_cached), calls_cached
            queries = ["{}\nknowledge triples: {}.".format(question, ". ".join(t)) for t in chain_texts]
            qe = aligner.calculate_query_embeddings(queries=queries, max_length=256).numpy()
            te = aligner.calculate_document_embeddings(documents=triples, max_length=128).numpy()
            so, io = S.search_canonical(qe, te, 20)
            assert np.array_equal(np.asarray(got[0], np.int64), io) and np.array_equal(np.asarray(got[1], np.float32).view(np.uint32), so.view(np.uint32))
            if hop == 2:                                                             # encoder parity at this hop's real shapes (2 x ~256 tokens, 512 triples)
                qr = _torch_fp32(enc, col, queries, "query", 256); tr = _torch_fp32(enc, col, triples, "doc", 128)
                ntok = int(col.encode_query(queries, max_length=256)["attention_mask"].sum(1).max())
                for out, ref in ((qe, qr), (te, tr)):
                    cos = (out * ref).sum(1) / (np.linalg.norm(out, axis=1) * np.linalg.norm(ref, axis=1))
                    assert np.abs(out - ref).max() <= 3e-3 and (1 - cos).max() <= 5e-5, (float(np.abs(out - ref).max()), float((1 - cos).max()))
                derr = float(np.abs(qe @ te.T - qr @ tr.T).max())
                print(f"[config5 {kind}] hop 2 encoder vs torch fp32: longest query {ntok} tokens, max |q.t - ref| = {derr:.2e}")
                assert derr <= 1e-3 and ntok >= 200
            # the chains grow by the best triple(s): 2 chains with longer texts on the next hop (query length -> 256 tokens)
            best = [triples[j] for j in got[0][0][:2]]
            chains = [chains[0] + best[:1] + all_triples[hop * 9: hop * 9 + 8], chains[0] + best[1:2] + all_triples[100 + hop * 9: 108 + hop * 9]]
            # ---- retrieval step: nq = 1 (hop 0) or 2, top-10 over the 5M rows ----------------------------------------------------------------------
            nq = 1 if hop == 0 else 2
            for qv in (planted[2 * hop: 2 * hop + nq].contiguous(), torch.from_numpy(qe[:nq]).to(dev).contiguous()):
                scans_before = ix.stats()["byte_scans"]
                s, i = ix.search(qv, 10)
                st = ix.stats()
                assert st["exact"] == 0, st
                # the hop's search really took the int8 pre-scan (a silently switched-off path - feedback state, a failed allocation -
                # would otherwise pass this test on the 16-bit round); the independent top-k below then checks what that path returned
                assert st["byte_scans"] == scans_before + 1 and st["byte_rows"] == N_ROWS, st
                searched_q.append(qv); searched.append((s, i))
            s_p, i_p = searched[-2]
            assert i_p[:, 0].tolist() == list(range(2 * hop, 2 * hop + nq))          # planted neighbour first
            print(f"[config5 {kind}] hop {hop}: nq={nq}, {n_tr} triples ({new_tr} new), search coarse {st['last_coarse_ms']:.2f} ms total {st['last_total_ms']:.2f} ms")
        aligner._embed = orig
    # ---- every searched query against the kernel-independent reference over the regenerated corpus ---------------------------------------------------
    allq = torch.cat(searched_q)
    all_s = np.concatenate([s for s, _ in searched]); all_i = np.concatenate([i for _, i in searched])
    g2 = torch.Generator(device=dev); g2.manual_seed(3)
    cd2 = CorpusDist(kind, D, dev)
    rs, ri = IC.torch_topk_fp32(allq, ((s0, cd2.rows(min(250_000, N_ROWS - s0), g2)) for s0 in range(0, N_ROWS, 250_000)), 10 + 32)
    out = IC.check_membership(all_s, all_i, rs.cpu().numpy(), ri.cpu().numpy(), 10)
    assert out["queries"] == len(allq) == 10
    xs = ix.reconstruct_rows(all_i.reshape(-1))
    sc = S.scores_at(allq.cpu().numpy(), xs, np.arange(all_i.size).reshape(all_i.shape).astype(np.int64))
    assert np.array_equal(sc.view(np.uint32), all_s.view(np.uint32))                 # canonical score bits of every returned row
    st = ix.stats()
    assert st["queries"] == 10 and st["certified"] + st["fine"] == 10, st
