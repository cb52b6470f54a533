"""GPU end-to-end tests of the drop-in surface: DenseRetriever / e5.py / compute_corpus_embeddings / faiss_index_corpus on the
HIP encoder + HIP index, checked against the reference-generated goldens (G8) and the oracle."""
import os
import pickle
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import encoder_np as E
from oracle import search_np as S

pytestmark = pytest.mark.gpu


def _setup(td, golden):
    from transformers import BertConfig
    from kirag_amd.bench_support import wordpiece_tokenizer
    from kirag_amd.retriever.encoders import E5Encoder
    g = golden("g4_g8_retriever.npz")
    H, L, heads, FF, vocab, max_pos = [int(v) for v in g["cfg"]]
    cfg = BertConfig(vocab_size=vocab, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=FF,
                     max_position_embeddings=max_pos)
    m = E5Encoder(cfg, add_pooling_layer=False)
    w = E.synth_weights(H, L, FF, vocab, max_pos, seed=int(g["weight_seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    m.save_pretrained(td)
    with open(os.path.join(td, "vocab.txt"), "w") as f:
        f.write("\n".join(str(v) for v in g["vocab"]) + "\n")
    tok = wordpiece_tokenizer(os.path.join(td, "vocab.txt"))
    return g, tok, w, heads


def test_dense_retriever_end_to_end(golden):
    from kirag_amd.collators import E5Collator
    from kirag_amd.retriever import e5 as e5mod
    from kirag_amd.retriever.index import Indexer
    from kirag_amd.retriever.retrievers import DenseRetriever, InBatchRetriever
    with tempfile.TemporaryDirectory() as td:
        g, tok, w, heads = _setup(td, golden)
        ret = InBatchRetriever("E5Retriever", td, temperature=0.01).cuda()
        col = E5Collator(tokenizer=tok, query_maxlength=16, doc_maxlength=24)
        queries = [str(v) for v in g["g7.queries"]]; docs = [str(v) for v in g["g7.docs"]]
        dr = DenseRetriever(retriever=ret, collator=col, indexer=None, corpus=None, batch_size=2)
        qe = dr.calculate_query_embeddings(queries); de = dr.calculate_document_embeddings(docs, max_length=16)
        assert qe.device.type == "cpu" and qe.dtype == torch.float32
        # G8: the reference encodes in batches of 2 (per-batch padding); rows are batch-invariant here
        assert np.abs(qe.numpy() - g["g8.qemb"]).max() <= 4e-3 and np.abs(de.numpy() - g["g8.demb"]).max() <= 4e-3
        # a 200-passage corpus through the whole surface
        rng = np.random.default_rng(1)
        words = [str(v) for v in g["vocab"] if str(v).isalpha() and len(str(v)) > 1]
        corpus_docs = ["title:  " + " ".join(rng.choice(words, 2)) + ", text:  " + " ".join(rng.choice(words, int(rng.integers(3, 18)))) for _ in range(200)]
        demb = dr.calculate_document_embeddings(corpus_docs)
        ix = Indexer(ret.hidden_size); ix.index_data([str(5000 + i) for i in range(200)], demb.numpy())
        dr.indexer = ix
        qs = ["who was born in paris", "capital of france river", corpus_docs[17].split("text:  ")[1]]
        out = dr(qs, topk=5)
        qv = dr.calculate_query_embeddings(qs).numpy()
        so, io = S.search_canonical(qv, demb.numpy(), 5)
        for r in range(3):
            assert [d["id"] for d in out[r]] == [str(5000 + j) for j in io[r]]
            assert np.array_equal(np.array([d["score"] for d in out[r]], np.float32), so[r])
        # encoder output vs the numpy oracle on the collator's own ids
        a = col.encode_doc(corpus_docs[:32])
        ref = E.e5_encode(w, a["input_ids"].numpy(), a["attention_mask"].numpy(), heads)
        assert np.abs(demb[:32].numpy() - ref).max() <= 4e-3
        # retriever/e5.py singleton sharing the resident encoder
        e5mod.set_model(ret.encoder, tok)
        eq = e5mod.get_e5_embeddings_for_query(queries, max_length=16)
        assert eq.device.type == "cpu" and np.abs(eq.numpy() - qe.numpy()).max() <= 1e-5
        ed = e5mod.get_e5_embeddings_for_document([d for d in docs], max_length=16)
        assert np.abs(ed.numpy() - de.numpy()).max() <= 1e-5


def test_corpus_encode_files_index_and_search(golden, tmp_path):
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd import faiss_index_corpus as FI
    from kirag_amd.collators import E5Collator
    from kirag_amd.retriever.index import Indexer
    from kirag_amd.retriever.retrievers import InBatchRetriever
    with tempfile.TemporaryDirectory() as td:
        g, tok, w, heads = _setup(td, golden)
        ret = InBatchRetriever("E5Retriever", td, temperature=0.01)
        col = E5Collator(tokenizer=tok, query_maxlength=16, doc_maxlength=32)
        rng = np.random.default_rng(2)
        words = [str(v) for v in g["vocab"] if str(v).isalpha() and len(str(v)) > 1]

        class Corpus:
            def __init__(self, n):
                self.p = ["title:  " + " ".join(rng.choice(words, 2)) + ", text:  " + " ".join(rng.choice(words, int(rng.integers(3, 25)))) for _ in range(n)]
                self.index_to_passage_id = {i: str(3 * i + 10) for i in range(n)}
            def __len__(self): return len(self.p)
            def __getitem__(self, i): return {"index": i, "passage": self.p[i]}
        corpus = Corpus(301)
        args = SimpleNamespace(local_rank=-1, save_dir=str(tmp_path), name="e5", index_folder="c", per_gpu_batch_size=8,
                               num_passage_per_index_file=100, encode_batch_size=64)
        resident = Indexer(ret.hidden_size)
        for r in range(2):      # two "ranks" run back to back on the one GPU
            CC.cal_doc_embeddings(args, ret, corpus, col, rank=r, world=2, indexer=resident if r == 0 else None)
        folder = os.path.join(str(tmp_path), "e5", "c")
        assert len([f for f in os.listdir(folder) if f.startswith("corpus_embeddings_")]) == 4
        built = FI.build_faiss_index(SimpleNamespace(index_folder=folder, embedding_size=ret.hidden_size))
        assert built.index.ntotal == 301 and not [f for f in os.listdir(folder) if f.endswith(".pkl")]
        assert sorted(os.listdir(folder)) == ["index.faiss", "index_meta.faiss"]
        assert built.index_id_to_db_id.tolist() == [3 * i + 10 for i in range(301)]
        loaded = Indexer(ret.hidden_size); loaded.deserialize_from(folder)
        x = loaded.index.reconstruct_n(0, 301)
        a = col.encode_doc(corpus.p[:40])
        ref = E.e5_encode(w, a["input_ids"].numpy(), a["attention_mask"].numpy(), heads)
        assert np.abs(x[:40] - ref).max() <= 4e-3
        assert np.array_equal(resident.index.reconstruct_n(0, 151), x[:151])          # rank 0's resident shard == its files
        q = x[[5, 150, 300]]
        res = loaded.search_knn(q, 3, verbose=False)
        so, io = S.search_canonical(q, x, 3)
        for r in range(3):
            assert res[r][0] == [str(3 * j + 10) for j in io[r]] and np.array_equal(res[r][1], so[r])


def test_setup_retriever_model_wiring(golden, tmp_path):
    """retrieve.py:86-118 on the HIP path: index folder written by the corpus-encode + index-build entry points, loaded back through
    setup_retriever_model, queried through DenseRetriever.__call__ with a corpus object; results equal the oracle on the stored embeddings."""
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd import faiss_index_corpus as FI
    from kirag_amd.collators import E5Collator
    from kirag_amd.retrieve import setup_retriever_model
    from kirag_amd.retriever import e5 as e5mod
    from kirag_amd.retriever.retrievers import InBatchRetriever
    with tempfile.TemporaryDirectory() as td:
        g, tok, w, heads = _setup(td, golden)
        rng = np.random.default_rng(3)
        words = [str(v) for v in g["vocab"] if str(v).isalpha() and len(str(v)) > 1]

        class Corpus:
            def __init__(self, n):
                self.p = ["title:  " + " ".join(rng.choice(words, 2)) + ", text:  " + " ".join(rng.choice(words, int(rng.integers(3, 25)))) for _ in range(n)]
                self.index_to_passage_id = {i: str(7 * i + 1) for i in range(n)}
            def __len__(self): return len(self.p)
            def __getitem__(self, i): return {"index": i, "passage": self.p[i]}
            def get_document(self, docid): return {"id": docid, "text": self.p[(int(docid) - 1) // 7]}
        corpus = Corpus(120)
        ret = InBatchRetriever("E5Retriever", td, temperature=0.01)
        col = E5Collator(tokenizer=tok, query_maxlength=16, doc_maxlength=32)
        enc_args = SimpleNamespace(local_rank=-1, save_dir=str(tmp_path), name="e5", index_folder="c", per_gpu_batch_size=8,
                                   num_passage_per_index_file=1000, encode_batch_size=64)
        CC.cal_doc_embeddings(enc_args, ret, corpus, col)
        folder = os.path.join(str(tmp_path), "e5", "c")
        FI.build_faiss_index(SimpleNamespace(index_folder=folder, embedding_size=ret.hidden_size))
        args = SimpleNamespace(retriever_name="E5Retriever", tokenizer_name_or_path=td, query_maxlength=16, doc_maxlength=32,
                               retriever_model_name_or_path=td, local_rank=-1, corpus="unused", index_folder=folder,
                               embedding_size=ret.hidden_size, per_gpu_batch_size=8)
        dr, cds = setup_retriever_model(args, corpus_dataset=corpus, tokenizer=tok)
        assert cds is corpus and dr.indexer.index.ntotal == 120 and e5mod.model is dr.retriever.encoder
        qs = [corpus.p[40].split("text:  ")[1], "capital of france"]
        out = dr(qs, topk=4)
        single = dr(qs[0], topk=4)
        assert single == out[0]
        x = dr.indexer.index.reconstruct_n(0, 120)
        so, io = S.search_canonical(dr.calculate_query_embeddings(qs).numpy(), x, 4)
        for r in range(2):
            assert [d["id"] for d in out[r]] == [str(7 * j + 1) for j in io[r]]
            assert all(isinstance(d["score"], float) for d in out[r]) and out[r][0]["text"] == corpus.p[io[r][0]]


def test_g4_g5_g6_on_the_gpu(golden):
    """SURVEY 8a rows a11-a13 on the GPU box against the reference-generated goldens:
    G4  BaseRetriever.compute_logits / score (retrievers.py:71-91), four rank cases + temperature 0.01 / "sqrt" + the ValueError, on device tensors;
    G5  rank-3 input_ids through get_encoder_output / encoder_embed / doc (retrievers.py:93-122) on the HIP encoder (eval mode);
    G6  InBatchRetriever.forward (retrievers.py:133-150): in train mode (PyTorch autograd on the GPU, loss.backward() reaches the weights) and in
        eval mode (HIP encoder) - loss / scores / query embeddings against the reference's."""
    from kirag_amd.retriever.retrievers import InBatchRetriever
    with tempfile.TemporaryDirectory() as td:
        g, tok, w, heads = _setup(td, golden)
        ret = InBatchRetriever("E5Retriever", td, temperature=0.01).cuda()
        assert ret.device.type == "cuda"
        T = lambda k: torch.from_numpy(g[k]).cuda()
        q1, d1, q2, d2, d3 = (T(f"g4.{k}") for k in ("q1", "d1", "q2", "d2", "d3"))
        for got, want, tol in ((ret.compute_logits(q1, d1), "g4.l11", 1e-6), (ret.compute_logits(q1, d2), "g4.l12", 1e-6),
                               (ret.compute_logits(q2, d3), "g4.l23", 1e-6), (ret.compute_logits(q2, d2), "g4.l22", 1e-6), (ret.score(q2, d2), "g4.s22_t001", 2e-4)):
            assert got.is_cuda
            np.testing.assert_allclose(got.cpu().numpy(), g[want], rtol=2e-6, atol=tol)
        ret.temperature = "sqrt"
        np.testing.assert_allclose(ret.score(q2, d2).cpu().numpy(), g["g4.s22_sqrt"], rtol=2e-6, atol=1e-6)
        ret.temperature = 0.01
        with pytest.raises(ValueError) as ei:
            ret.compute_logits(d3, d3)
        assert str(ei.value) == str(g["g4.err"])
        # G5: rank-3 ids through the HIP encoder
        ret.eval()
        a3 = {"input_ids": T("g5.ids"), "attention_mask": T("g5.mask")}
        out3 = ret.doc(a3)
        assert out3.is_cuda and tuple(out3.shape) == tuple(g["g5.out"].shape)
        assert np.abs(out3.cpu().numpy() - g["g5.out"]).max() <= 4e-3
        # G6 eval mode (HIP forward) and train mode (autograd)
        qa = {"input_ids": T("g7.e5.q.ids"), "attention_mask": T("g7.e5.q.mask")}
        da = {"input_ids": T("g7.e5.d.ids"), "attention_mask": T("g7.e5.d.mask")}
        labels = torch.tensor([0, 1, 2], device="cuda")
        with torch.no_grad():
            loss_e, scores_e, gq_e, gd_e = ret(qa, da, labels)
        assert np.abs(gq_e.cpu().numpy() - g["g6.q"]).max() <= 4e-3
        np.testing.assert_allclose(scores_e.cpu().numpy(), g["g6.scores"], atol=0.1)          # scores / 0.01: 1e-3 on a cosine = 0.1 here
        assert abs(float(loss_e) - float(g["g6.loss"])) < 5e-2
        ret.train()
        for mod in ret.modules():                       # the golden was taken without dropout (config dropout 0 in make_golden.py)
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        loss, scores, gq, gd = ret(qa, da, labels)
        assert loss.requires_grad and loss.is_cuda
        np.testing.assert_allclose(gq.detach().cpu().numpy(), g["g6.q"], atol=5e-5)
        np.testing.assert_allclose(scores.detach().cpu().numpy(), g["g6.scores"], atol=2e-2)
        assert abs(float(loss) - float(g["g6.loss"])) < 5e-3
        loss.backward()
        gw = ret.encoder.encoder.layer[0].attention.self.query.weight.grad
        assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().sum()) > 0


def test_config4_flow_streamed_bge_encode_into_resident_shard_then_search(golden, tmp_path):
    """BASELINE config 4's flow (bge-large 'streamed encode + search') at test size: 200k synthetic passages through BGECollator (no passage
    prefix) -> BGE encoder (CLS pooling) on the HIP path -> straight into a resident ShardedIndexer shard (no embedding files, no host round trip
    per batch: compute_corpus_embeddings.py:77-125 without the gather / torch.cat), then queries through BGECollator.encode_query
    (instruction prefix) -> search_knn.  Checks: every passage indexed exactly once and in corpus order, stored rows == a direct batch of the
    same passages (batch invariance) == the numpy oracle's BGE forward (<= 4e-3), search results == the C oracle on the stored rows, native
    shard round trip; both tokenisation feeds (prefetch thread, worker processes) give identical rows."""
    import time
    from transformers import BertConfig
    from kirag_amd.bench_support import wordpiece_tokenizer
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd.collators import BGECollator
    from kirag_amd.retriever.encoders import BGEEncoder
    from kirag_amd.retriever.index import ShardedIndexer
    from kirag_amd.retriever.retrievers import InBatchRetriever
    with tempfile.TemporaryDirectory() as td:
        g = golden("g4_g8_retriever.npz")
        H, L, heads, FF, vocab, max_pos = [int(v) for v in g["cfg"]]
        cfg = BertConfig(vocab_size=vocab, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=FF, max_position_embeddings=max_pos)
        m = BGEEncoder(cfg, add_pooling_layer=False)
        w = E.synth_weights(H, L, FF, vocab, max_pos, seed=int(g["weight_seed"]))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
        m.save_pretrained(td)
        with open(os.path.join(td, "vocab.txt"), "w") as f:
            f.write("\n".join(str(v) for v in g["vocab"]) + "\n")
        tok = wordpiece_tokenizer(os.path.join(td, "vocab.txt"))
        ret = InBatchRetriever("BGERetriever", td, temperature=0.01)
        col = BGECollator(tokenizer=tok, query_maxlength=32, doc_maxlength=32)
        rng = np.random.default_rng(5)
        words = np.array([str(v) for v in g["vocab"] if str(v).isalpha() and len(str(v)) > 1])
        n = 200_000
        lens = rng.integers(4, 26, n)
        flat = rng.choice(words, int(lens.sum()))
        cuts = np.concatenate([[0], np.cumsum(lens)])

        class Corpus:
            index_to_passage_id = {i: str(2 * i + 7) for i in range(n)}
            def __len__(self): return n
            def passage(self, i): return "title:  " + flat[cuts[i]] + ", text:  " + " ".join(flat[cuts[i] + 1:cuts[i + 1]])
            def __getitem__(self, i): return {"index": i, "passage": self.passage(i)}
        corpus = Corpus()
        args = SimpleNamespace(local_rank=-1, save_dir=str(tmp_path), name="bge", index_folder="c", per_gpu_batch_size=8, num_passage_per_index_file=1_000_000,
                               encode_batch_size=2048, prefetch_batches=4, tokenizer_workers=0, no_embedding_files=True)
        resident = ShardedIndexer(H)
        t0 = time.perf_counter()
        CC.cal_doc_embeddings(args, ret, corpus, col, indexer=resident)
        dt = time.perf_counter() - t0
        print(f"[config-4 flow] {n} passages streamed into the resident shard in {dt:.1f} s = {n / dt:.0f} passages/s (tiny encoder: the tokenizer feed is the limit)")
        assert resident.index.ntotal == n and not [f for f in os.listdir(os.path.join(str(tmp_path), "bge", "c")) if f.endswith(".pkl")]
        qtexts = [" ".join(flat[cuts[i] + 1:cuts[i + 1]]) for i in (5, 70_000, 199_999)] + ["capital of france"]
        qa = col.encode_query(qtexts)
        qv = ret.query({k: v.cuda() for k, v in qa.items()}).cpu().numpy()
        res = resident.search_knn(qv, 10, verbose=False)
        assert resident.ntotal_global == n and resident.index_id_to_db_id[:3].tolist() == [7, 9, 11] and resident.index_id_to_db_id[-1] == 2 * (n - 1) + 7
        x = resident.index.reconstruct_n(0, n)
        so, io = S.search_canonical(qv, x, 10)
        for r in range(4):
            assert res[r][0] == [str(2 * j + 7) for j in io[r]] and np.array_equal(res[r][1].view(np.uint32), so[r].view(np.uint32))
        pick = rng.choice(n, 64, replace=False)
        a = col.encode_doc([corpus.passage(int(i)) for i in pick])
        direct = ret.doc({k: v.cuda() for k, v in a.items()}).cpu().numpy()
        assert np.array_equal(direct, x[pick])                                                        # batch invariance: bit-equal rows
        ref = E.bge_encode(w, a["input_ids"].numpy(), a["attention_mask"].numpy(), heads)
        assert np.abs(direct - ref).max() <= 4e-3
        # the same stream tokenised by worker processes, a smaller slice: identical rows
        args2 = SimpleNamespace(**{**vars(args), "tokenizer_workers": 4})
        small = type("C2", (), {"index_to_passage_id": corpus.index_to_passage_id, "__len__": lambda s: 20_000, "__getitem__": lambda s, i: corpus[i]})()
        res2 = ShardedIndexer(H)
        CC.cal_doc_embeddings(args2, ret, small, col, indexer=res2)
        assert res2.index.ntotal == 20_000 and np.array_equal(res2.index.reconstruct_n(0, 20_000), x[:20_000])
        nat = os.path.join(str(tmp_path), "native")
        resident.serialize(nat)
        back = ShardedIndexer(H); back.deserialize_from(nat)
        res3 = back.search_knn(qv, 10, verbose=False)
        assert all(res3[r][0] == res[r][0] and np.array_equal(res3[r][1], res[r][1]) for r in range(4))


def test_token_type_ids_go_through_the_hip_path(golden):
    """HF BertModel.forward's third input.  No KiRAG caller passes non-zero token types (the collators encode single texts), but E5Encoder / BGEEncoder.forward have
    the argument (encoders.py:67,106); until round 4 the HIP path refused non-zero values.  Now they reach the embedding kernel (kr_encoder_forward_tt): checked
    against the numpy oracle's bert_forward and against the module's own PyTorch forward; host and device tensors alike, no host synchronisation on the forward;
    a value outside the type vocabulary is a deferred KR_EINVAL like a token id outside the vocabulary."""
    from transformers import BertConfig
    from kirag_amd import _lib
    from kirag_amd.retriever.encoders import E5Encoder
    from oracle import encoder_np as E
    g = golden("g4_g8_retriever.npz")
    H, L, heads, FF, vocab, max_pos = [int(v) for v in g["cfg"]]
    cfg = BertConfig(vocab_size=vocab, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=FF, max_position_embeddings=max_pos)
    torch.manual_seed(3)
    m = E5Encoder(cfg, add_pooling_layer=False).cuda().eval()
    with torch.no_grad():
        m.embeddings.token_type_embeddings.weight.mul_(25.0)    # make the two type rows matter (HF initialises them at N(0, 0.02))
    rng = np.random.default_rng(5)
    ids = torch.from_numpy(rng.integers(5, vocab, (6, 20))); mask = torch.ones_like(ids); mask[1, 13:] = 0; mask[4, :7] = 0
    tt = torch.from_numpy(rng.integers(0, 2, (6, 20)))
    zeros = torch.zeros_like(ids)
    out_tt = m(ids.cuda(), mask.cuda(), tt.cuda())
    out_0 = m(ids.cuda(), mask.cuda(), zeros.cuda())
    m._hip.check()
    assert torch.equal(m(ids.cuda(), mask.cuda(), tt), out_tt)                       # a host tensor of types gives the same bits
    assert torch.equal(m(ids.cuda(), mask.cuda()), out_0) and torch.equal(m(ids.cuda(), mask.cuda(), zeros), out_0)
    assert (out_tt - out_0).abs().max() > 1e-2                                       # the types are not ignored
    # the numpy oracle (BertModel semantics, retriever/encoders.py:67-77)
    w = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
    hid = E.bert_forward(w, ids.numpy(), mask.numpy(), heads, token_type_ids=tt.numpy())
    ref = E.average_pool(hid, mask.numpy()); ref = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    assert np.abs(out_tt.cpu().numpy() - ref).max() < 4e-3
    # the module's own PyTorch forward (what train mode runs)
    with torch.no_grad():
        lh = m._torch_pooled(ids.cuda(), mask.cuda(), tt.cuda())
    mm = mask.cuda()[..., None].float()
    tref = torch.nn.functional.normalize((lh * mm).sum(1) / mm.sum(1), dim=1)
    assert (out_tt - tref).abs().max() < 4e-3
    # a type outside [0, type_vocab_size): the forward only enqueues, check() reports it once, the handle stays usable
    bad = tt.clone(); bad[2, 3] = 7
    out = m(ids.cuda(), mask.cuda(), bad.cuda())
    assert out.shape == out_tt.shape
    with pytest.raises(_lib.KiragAmdError, match="token_type_ids"):
        m._hip.check()
    m._hip.check()
    assert torch.equal(m(ids.cuda(), mask.cuda(), tt.cuda()), out_tt)
    m._hip.check()


def test_module_surface_weight_sync_is_cached_and_still_sees_every_change():
    """Round 5: ``HipBertForward.sync`` used to walk ``named_parameters()`` on EVERY eval forward — ~0.7 ms of host time for a 24-layer BertModel, as
    long as the 32-token forward behind it.  The walk is cached now (per forward: the tensor versions of the cached list).  Every way the weights can
    change must still reach the library: in-place updates (version bump), ``load_state_dict``, dtype / device moves through ``_apply``, a parameter object
    replaced by assignment (``invalidate_hip_weights()`` at once, the periodic full check at the latest), train / eval transitions."""
    import time
    from transformers import BertConfig
    from kirag_amd.retriever.encoders import E5Encoder
    cfg = BertConfig(vocab_size=500, hidden_size=128, num_hidden_layers=24, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)       # the PyTorch reference below runs in train mode: no dropout
    torch.manual_seed(7)
    m = E5Encoder(cfg, add_pooling_layer=False).cuda().eval()
    rng = np.random.default_rng(1)
    ids = torch.from_numpy(rng.integers(5, 500, (3, 17))).cuda(); mask = torch.ones_like(ids); mask[1, 9:] = 0

    def hip(): return m(ids, mask).clone()

    def ref():
        m.train()
        try:
            with torch.no_grad():
                return m(ids, mask).clone()
        finally:
            m.eval()
    out0 = hip()
    assert (out0 - ref()).abs().max() < 4e-3
    # the steady state: the cached check is cheap (the full walk of 24 layers is not)
    m._hip.sync(m)                                                  # ref() went through train() / eval(): this call re-reads the weights once
    t0 = time.perf_counter()
    for _ in range(50): m._hip.sync(m)
    cached_us = (time.perf_counter() - t0) / 50 * 1e6
    t0 = time.perf_counter()
    for _ in range(10): list(m.named_parameters())
    walk_us = (time.perf_counter() - t0) / 10 * 1e6
    print(f"sync with the cached walk {cached_us:.0f} us per forward; one named_parameters() walk {walk_us:.0f} us")
    assert cached_us < 0.5 * walk_us
    m._hip.invalidate(); assert torch.equal(hip(), out0)          # after an invalidate: the same weights, the same bits
    # 1. in-place update (bumps the tensor version)
    with torch.no_grad():
        m.encoder.layer[5].output.dense.weight.mul_(1.7)
    out1 = hip()
    assert (out1 - out0).abs().max() > 1e-4 and (out1 - ref()).abs().max() < 4e-3
    # 2. load_state_dict back to the old values -> the old bits
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        m.encoder.layer[5].output.dense.weight.div_(1.7)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd); assert torch.equal(hip(), out1)
    m.load_state_dict(sd0); assert (hip() - out0).abs().max() < 1e-5
    # 3. _apply: .half().float() rounds the weights -> different outputs, equal to the module's own forward
    m.half().float()
    out3 = hip()
    assert (out3 - ref()).abs().max() < 4e-3 and not torch.equal(out3, out0)
    # 4. a parameter object replaced by assignment: noticed at once after invalidate_hip_weights(), and by the periodic full check without it
    #    (replacements large enough that stale weights could not pass the comparison with the module's own forward)
    w_old = m.encoder.layer[3].intermediate.dense.weight.detach()
    m.encoder.layer[3].intermediate.dense.weight = torch.nn.Parameter(torch.flip(w_old, dims=[0]) * 3.0)
    m.invalidate_hip_weights()
    out4 = hip()
    r4 = ref()
    assert (r4 - out3).abs().max() > 2e-2, float((r4 - out3).abs().max())        # the replacement matters ...
    assert (out4 - r4).abs().max() < 4e-3                                          # ... and the HIP path has it
    m._hip.FULL_CHECK_EVERY = 3
    w_old = m.encoder.layer[9].output.dense.weight.detach()
    m.encoder.layer[9].output.dense.weight = torch.nn.Parameter(torch.flip(w_old, dims=[1]) * 3.0)
    outs = [hip() for _ in range(4)]                                # no invalidate: the periodic full walk finds the new object
    r5 = ref()
    assert (r5 - r4).abs().max() > 2e-2 and (outs[-1] - r5).abs().max() < 4e-3
