"""CPU-side checks: the C-ABI library loads and exports every symbol include/kirag_amd.h declares, fails loudly without a
GPU (no fallback), host-only entry points work, and the host-side mirrors of the reference surface behave as pinned by the
golden vectors."""
import ctypes as C
import os
import pickle
import re
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from kirag_amd import _lib
from oracle import search_np as S

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = not torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "kirag_amd.h")).read()
    declared = set(re.findall(r"\b(kr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 19
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.kr_abi_version() == _lib.ABI_VERSION == 9


def test_search_stats_struct_and_options_mirror_the_header():
    """ABI 8: the ctypes mirror of kr_search_stats has the header's fields in the header's order (two new ones at the end: the byte pre-scan of small
    query blocks), every option name the header documents is accepted, an unknown one is KR_EINVAL."""
    hdr = open(os.path.join(REPO, "include", "kirag_amd.h")).read()
    end = hdr.index("} kr_search_stats;")
    body = hdr[hdr.rindex("typedef struct {", 0, end):end]
    fields = re.findall(r"^\s*(?:int64_t|double)\s+([a-z0-9_]+)\s*;", body, flags=re.M)
    assert fields == [f for f, _ in _lib.SearchStats._fields_] and fields[-3:] == ["byte_scans", "byte_marked_rows", "byte_rows"]
    assert C.sizeof(_lib.SearchStats) == 8 * len(fields)
    lib = _lib.load()
    for name, back in ((b"byte_prescan", 1), (b"debug_byte_min_rows", -1), (b"force_exact_scores", 0), (b"debug_eps8_permille", 1000)):
        assert name.decode() in hdr
        assert lib.kr_set_option(name, 0) == 0 and lib.kr_set_option(name, back) == 0
    assert lib.kr_set_option(b"no_such_option", 1) == -22


@pytest.mark.skipif(not NO_GPU, reason="checks the no-GPU failure mode")
def test_compute_calls_fail_loudly_without_gpu():
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.kr_index_create(64, 0, 0, 0, C.byref(h))
    assert rc == -19 and b"no CPU fallback" in lib.kr_last_error()
    from kirag_amd.retriever.index import Indexer
    with pytest.raises(_lib.KiragAmdError):
        Indexer(64)
    cfg = _lib.BertCfg(128, 2, 2, 512, 100, 64, 2, 1e-12)
    assert lib.kr_encoder_create(C.byref(cfg), 0, C.byref(h)) == -19


def test_argument_validation_before_any_device_work():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.kr_index_create(63, 0, 0, 0, C.byref(h)) == -22 and b"vector size" in lib.kr_last_error()
    assert lib.kr_index_create(64, 1, 0, 0, C.byref(h)) == -22          # l2 metric not implemented
    cfg = _lib.BertCfg(100, 2, 2, 512, 100, 64, 2, 1e-12)               # hidden not a multiple of 128
    assert lib.kr_encoder_create(C.byref(cfg), 0, C.byref(h)) == -22
    from kirag_amd.retriever.index import Indexer
    with pytest.raises(NotImplementedError):
        Indexer(64, metric="l2")
    with pytest.raises(NotImplementedError):
        Indexer(64, n_subquantizers=8)


def test_host_merge_matches_oracle():
    from kirag_amd.parallel import merge_topk
    rng = np.random.default_rng(0)
    x = rng.standard_normal((900, 32)).astype(np.float32); q = rng.standard_normal((11, 32)).astype(np.float32)
    x[700] = x[20]
    so, io = S.search_canonical(q, x, 16)
    ss, ii = [], []
    for a, b in ((0, 300), (300, 310), (310, 900)):              # middle shard shorter than k -> padded with id -1
        kk = min(16, b - a)
        s_, i_ = S.search_canonical(q, x[a:b], kk)
        s_ = np.pad(s_, ((0, 0), (0, 16 - kk)), constant_values=-np.inf); i_ = np.pad(i_ + a, ((0, 0), (0, 16 - kk)), constant_values=-1)
        ss.append(s_); ii.append(i_)
    ms, mi = merge_topk(np.stack(ss), np.stack(ii), 16)
    assert np.array_equal(mi, io) and np.array_equal(ms, so)


def test_host_merge_threaded_path_matches_numpy():
    """8 shards x 600 queries x 50: large enough for kr_topk_merge to split the queries over several host threads; compared with a lexsort of the
    concatenated lists (score desc, id asc), with duplicated scores across shards."""
    from kirag_amd.parallel import merge_topk
    rng = np.random.default_rng(5)
    W, nq, k = 8, 600, 50
    sc = np.round(rng.standard_normal((W, nq, k)).astype(np.float32), 1)          # coarse grid -> many ties between shards
    ids = np.stack([rng.permuted(np.tile(np.arange(w * 10_000, w * 10_000 + 5_000), (nq, 1)), axis=1)[:, :k] for w in range(W)]).astype(np.int64)
    order = np.lexsort((ids, -sc), axis=2)                                         # every shard list sorted by (score desc, id asc)
    sc = np.take_along_axis(sc, order, 2); ids = np.take_along_axis(ids, order, 2)
    ms, mi = merge_topk(sc, ids, k)
    cs = np.transpose(sc, (1, 0, 2)).reshape(nq, W * k); ci = np.transpose(ids, (1, 0, 2)).reshape(nq, W * k)
    o = np.lexsort((ci, -cs), axis=1)[:, :k]
    assert np.array_equal(mi, np.take_along_axis(ci, o, 1)) and np.array_equal(ms, np.take_along_axis(cs, o, 1))


def _tiny_model_dir(td, golden):
    from transformers import BertConfig
    from kirag_amd.retriever.encoders import E5Encoder
    from oracle import encoder_np as E
    g = golden("g4_g8_retriever.npz")
    H, L, heads, FF, vocab, max_pos = [int(v) for v in g["cfg"]]
    cfg = BertConfig(vocab_size=vocab, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=FF,
                     max_position_embeddings=max_pos, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = E5Encoder(cfg, add_pooling_layer=False)
    w = E.synth_weights(H, L, FF, vocab, max_pos, seed=int(g["weight_seed"]))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    m.save_pretrained(td)
    with open(os.path.join(td, "vocab.txt"), "w") as f:
        f.write("\n".join(str(v) for v in g["vocab"]) + "\n")
    return g


def test_collators_match_reference_golden(golden):
    from kirag_amd.bench_support import wordpiece_tokenizer
    from kirag_amd.collators import COLLATOR_MAP, BGECollator, E5Collator
    g = golden("g4_g8_retriever.npz")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "vocab.txt"), "w") as f:
            f.write("\n".join(str(v) for v in g["vocab"]) + "\n")
        tok = wordpiece_tokenizer(os.path.join(td, "vocab.txt"))
    queries = [str(v) for v in g["g7.queries"]]; docs = [str(v) for v in g["g7.docs"]]
    for nm, col in (("e5", E5Collator(tok, 16, 24)), ("bge", BGECollator(tok, 24, 24))):
        qa, da, q8 = col.encode_query(queries), col.encode_doc(docs), col.encode_query(queries, max_length=8)
        assert np.array_equal(qa["input_ids"].numpy(), g[f"g7.{nm}.q.ids"]) and np.array_equal(qa["attention_mask"].numpy(), g[f"g7.{nm}.q.mask"])
        assert np.array_equal(da["input_ids"].numpy(), g[f"g7.{nm}.d.ids"]) and np.array_equal(da["attention_mask"].numpy(), g[f"g7.{nm}.d.mask"])
        assert np.array_equal(q8["input_ids"].numpy(), g[f"g7.{nm}.q8.ids"])
    assert set(COLLATOR_MAP) == {"E5Retriever", "BGERetriever"}
    with pytest.raises(ValueError):
        E5Collator(tok, 16).encode_query([])


def test_retriever_registry_logits_score_and_train_path(golden):
    """RETRIEVER_MAP / load_retriever / BaseRetriever.compute_logits / score (G4), the training (autograd) forward of
    InBatchRetriever against the reference's loss and scores (G6), and the loud failure of eval mode on CPU."""
    from kirag_amd.retriever import retrievers as R
    assert set(R.RETRIEVER_MAP) == {"E5Retriever", "BGERetriever"}
    with pytest.raises(KeyError):
        R.load_retriever("Contriever", "x")
    with tempfile.TemporaryDirectory() as td:
        g = _tiny_model_dir(td, golden)
        ret = R.InBatchRetriever("E5Retriever", td, temperature=0.01)
        assert ret.hidden_size == int(g["cfg"][0]) and ret.world_size == 1 and ret.device.type == "cpu"
        T = torch.from_numpy
        q1, d1, q2, d2, d3 = (T(g[f"g4.{k}"]) for k in ("q1", "d1", "q2", "d2", "d3"))
        np.testing.assert_allclose(ret.compute_logits(q1, d1).numpy(), g["g4.l11"], rtol=1e-6)
        np.testing.assert_allclose(ret.compute_logits(q1, d2).numpy(), g["g4.l12"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ret.compute_logits(q2, d3).numpy(), g["g4.l23"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ret.compute_logits(q2, d2).numpy(), g["g4.l22"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ret.score(q2, d2).numpy(), g["g4.s22_t001"], rtol=1e-6, atol=1e-4)
        ret.temperature = "sqrt"
        np.testing.assert_allclose(ret.score(q2, d2).numpy(), g["g4.s22_sqrt"], rtol=1e-6, atol=1e-6)
        ret.temperature = 0.01
        with pytest.raises(ValueError) as ei:
            ret.compute_logits(d3, d3)
        assert str(ei.value) == str(g["g4.err"])
        qa = {"input_ids": T(g["g7.e5.q.ids"]), "attention_mask": T(g["g7.e5.q.mask"])}
        da = {"input_ids": T(g["g7.e5.d.ids"]), "attention_mask": T(g["g7.e5.d.mask"])}
        ret.train()                                              # training = inherited PyTorch autograd path (CPU ok)
        loss, scores, gq, gd = ret(qa, da, torch.tensor([0, 1, 2]))
        assert loss.requires_grad
        np.testing.assert_allclose(gq.detach().numpy(), g["g6.q"], atol=2e-5)
        np.testing.assert_allclose(scores.detach().numpy(), g["g6.scores"], atol=5e-3)
        assert abs(float(loss) - float(g["g6.loss"])) < 5e-3
        a3 = {"input_ids": T(g["g5.ids"]), "attention_mask": T(g["g5.mask"])}
        np.testing.assert_allclose(ret.doc(a3).detach().numpy(), g["g5.out"], atol=2e-5)       # G5: rank-3 input_ids
        ret.eval()
        if NO_GPU:
            with pytest.raises(RuntimeError) as ei:
                ret.query(qa)
            assert "no CPU fallback" in str(ei.value)
        ret.save_model(os.path.join(td, "resaved")); ret.load_model(os.path.join(td, "resaved"))
        assert type(ret.encoder).__name__ == "E5Encoder"


class _FakeRetriever(torch.nn.Module):
    """Deterministic stand-in encoder for host-logic tests: embedding = normalised bag of token ids."""
    def __init__(self, d=16):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(200, d, generator=torch.Generator().manual_seed(0)))
    @property
    def device(self):
        return self.w.device
    def _e(self, a):
        m = a["attention_mask"].unsqueeze(-1).float()
        return torch.nn.functional.normalize((self.w[a["input_ids"] % 200] * m).sum(1), dim=-1)
    def query(self, a, **k): return self._e(a)
    def doc(self, a, **k): return self._e(a)


class _WordTok:
    """whitespace tokenizer with padding=True / truncation semantics, enough for collator + DenseRetriever plumbing"""
    pad_token = "[PAD]"; pad_token_id = 0
    def __call__(self, texts, max_length, padding, truncation, return_tensors):
        rows = [[(hash(w) % 190) + 5 for w in t.split()][:max_length] for t in texts]
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.zeros(len(rows), L, dtype=torch.long); mask = torch.zeros_like(ids)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r); mask[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}


class _OracleBackedIndexer(S.OracleIndexer):
    pass


def test_dense_retriever_result_structure_and_batch_invariance():
    from kirag_amd.collators import E5Collator
    from kirag_amd.retriever.retrievers import DenseRetriever
    ret = _FakeRetriever(); col = E5Collator(_WordTok(), 12, 20)
    docs = [f"title:  t{i}, text:  word{i} common tail {i % 7}" for i in range(60)]
    ix = _OracleBackedIndexer(16)
    dr = DenseRetriever(ret, col, indexer=None, corpus=None, batch_size=4, encode_batch_size=8)
    demb = dr.calculate_document_embeddings(docs)
    assert demb.device.type == "cpu" and tuple(demb.shape) == (60, 16)
    dr2 = DenseRetriever(ret, col, batch_size=4, encode_batch_size=64)
    torch.testing.assert_close(dr2.calculate_document_embeddings(docs), demb)          # chunking does not change rows
    ix.index_data([str(1000 + i) for i in range(60)], demb.numpy())
    with pytest.raises(AssertionError):
        dr("q", 3)                                                                       # indexer missing
    with pytest.raises(AssertionError):
        dr.calculate_query_embeddings([])
    dr.indexer = ix
    out = dr(["word3 common", "word10"], topk=4)
    assert len(out) == 2 and len(out[0]) == 4
    assert set(out[0][0]) == {"id", "score"} and isinstance(out[0][0]["id"], str) and isinstance(out[0][0]["score"], np.float32)
    one = dr("word3 common", topk=4)
    assert [d["id"] for d in one] == [d["id"] for d in out[0]]

    class Corpus:
        def get_document(self, pid): return {"id": pid, "title": "t", "text": "x"}
    dr.corpus = Corpus()
    out = dr(["word3 common"], topk=2)
    assert isinstance(out[0][0]["score"], float) and out[0][0]["title"] == "t"
    docs_sorted = dr.get_documents({"1003": 0.1, "1004": 0.9})
    assert [d["id"] for d in docs_sorted] == ["1004", "1003"] and docs_sorted[0]["score"] == 0.9
    with pytest.raises(ValueError):
        dr.get_documents(("1003",))


def test_corpus_embedding_shard_files_and_index_builder(tmp_path):
    """cal_doc_embeddings' file contract (compute_corpus_embeddings.py:101-120) with a fake encoder on CPU, then the
    faiss_index_corpus reader's ordering / id matching (faiss_index_corpus.py:23-52) up to the Indexer boundary."""
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd.collators import E5Collator

    class Corpus:
        def __init__(self, n):
            self.index_to_passage_id = {i: str(7 * i + 1) for i in range(n)}
        def __len__(self): return len(self.index_to_passage_id)
        def __getitem__(self, i): return {"index": i, "passage": f"title:  t{i}, text:  word{i} tail {i % 5}"}

    corpus = Corpus(53); ret = _FakeRetriever(); col = E5Collator(_WordTok(), 12, 20)
    args = SimpleNamespace(local_rank=-1, save_dir=str(tmp_path), name="n", index_folder="f", per_gpu_batch_size=8,
                           num_passage_per_index_file=20, encode_batch_size=16)
    spans = [CC.cal_doc_embeddings(args, ret, corpus, col, rank=r, world=2, device=torch.device("cpu")) for r in range(2)]
    assert spans == [(0, 27), (27, 53)] and CC.shard_range(53, 1, 2) == (27, 53) and CC.shard_range(5, 7, 8) == (5, 5)
    folder = os.path.join(str(tmp_path), "n", "f")
    files = sorted(os.listdir(folder))
    assert "corpus_embeddings_0_19.pkl" in files and "corpus_embeddings_20_26.pkl" in files and "passage_id_list_47_52.pkl" in files
    from kirag_amd.faiss_index_corpus import sort_embedding_files
    import glob
    embs, ids = [], []
    for f in sort_embedding_files(glob.glob(os.path.join(folder, "corpus_embeddings_*.pkl"))):
        e = pickle.load(open(f, "rb")); embs.append(e)
        ids += pickle.load(open(f.replace("corpus_embeddings", "passage_id_list"), "rb"))
        assert isinstance(e, torch.Tensor) and e.dtype == torch.float32 and len(e) <= 20
    full = torch.cat(embs)
    assert ids == [str(7 * i + 1) for i in range(53)]
    ref = ret.doc(col.encode_doc([corpus[i]["passage"] for i in range(53)]))
    torch.testing.assert_close(full, ref.detach())


def test_corpus_encode_rejects_a_bad_batch_before_it_reaches_files_or_the_shard(tmp_path):
    """ADVICE r02 / VERDICT r03 item 6: a token id outside the vocabulary used to be reported only by enc.check() after the LAST flush, when the bad
    rows already sat in the resident shard and in a .pkl file.  Now the host validates every batch before it is sent anywhere
    (compute_corpus_embeddings.py: validate)."""
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd.collators import E5Collator

    class Corpus:
        def __init__(self, n):
            self.index_to_passage_id = {i: str(i) for i in range(n)}
        def __len__(self): return len(self.index_to_passage_id)
        def __getitem__(self, i): return {"index": i, "passage": f"title:  t{i}, text:  word{i}" + (" BAD" if i == 37 else "")}

    class Tok(_WordTok):
        def __call__(self, texts, **kw):
            out = super().__call__(texts, **kw)
            for r, t in enumerate(texts):
                if "BAD" in t:
                    out["input_ids"][r, 1] = 999_999        # outside the 200-entry vocabulary
            return out

    class Shard:                                            # the indexer surface cal_doc_embeddings uses
        def __init__(self): self.ids = []
        def index_data(self, ids, emb): self.ids += list(ids)

    ret = _FakeRetriever(); ret.encoder = SimpleNamespace(config=SimpleNamespace(vocab_size=200))
    col = E5Collator(Tok(), 12, 20)
    args = SimpleNamespace(local_rank=-1, save_dir=str(tmp_path), name="n", index_folder="f", per_gpu_batch_size=8, num_passage_per_index_file=16,
                           encode_batch_size=16)
    shard = Shard()
    with pytest.raises(ValueError, match="outside"):
        CC.cal_doc_embeddings(args, ret, Corpus(53), col, rank=0, world=1, device=torch.device("cpu"), indexer=shard)
    assert shard.ids == [str(i) for i in range(32)]        # the two good batches before it; not one row of the bad batch (passages 32..47)
    folder = os.path.join(str(tmp_path), "n", "f")
    for f in os.listdir(folder):                            # whatever was flushed holds good rows only
        if f.startswith("passage_id_list_"):
            assert all(int(i) < 32 for i in pickle.load(open(os.path.join(folder, f), "rb")))


def test_corpus_encode_accepts_an_out_of_vocabulary_id_at_a_masked_position(tmp_path):
    """the host-side check looked at ALL positions of input_ids, but the HIP forward only reads attended ones, and __main__ adds a '[PAD]'
    token with id == len(tokenizer) >= vocab_size when the tokenizer has none — every padded batch was rejected.  Only attended ids are validated now."""
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd.collators import E5Collator

    class Corpus:
        def __init__(self, n):
            self.index_to_passage_id = {i: str(i) for i in range(n)}
        def __len__(self): return len(self.index_to_passage_id)
        def __getitem__(self, i): return {"index": i, "passage": f"title:  t{i}, text:  word{i}" + (" some more words" if i % 5 == 0 else "")}

    class Tok(_WordTok):
        pad_token_id = 200                                   # == vocab_size: a pad token appended to the tokenizer after the model was trained
        def __call__(self, texts, **kw):
            out = super().__call__(texts, **kw)
            out["input_ids"][out["attention_mask"] == 0] = 200
            return out

    class Shard:
        def __init__(self): self.ids = []
        def index_data(self, ids, emb): self.ids += list(ids)

    ret = _FakeRetriever(); ret.encoder = SimpleNamespace(config=SimpleNamespace(vocab_size=200))
    args = SimpleNamespace(local_rank=-1, save_dir=str(tmp_path), name="n", index_folder="f", per_gpu_batch_size=8, num_passage_per_index_file=16,
                           encode_batch_size=16)
    shard = Shard()
    CC.cal_doc_embeddings(args, ret, Corpus(40), E5Collator(Tok(), 12, 20), rank=0, world=1, device=torch.device("cpu"), indexer=shard)
    assert shard.ids == [str(i) for i in range(40)]         # every (padded) batch went through


def test_faiss_flat_file_layout_byte_for_byte(tmp_path):
    """index.faiss as faiss 1.8 writes an IndexFlatIP (impl/index_write.cpp: fourcc "IxFI"; write_index_header = int d, idx_t ntotal, two idx_t
    dummies of 1 << 20, bool is_trained, int metric_type (0 = METRIC_INNER_PRODUCT, no metric_arg); then WRITEXBVECTOR(codes): size_t count =
    bytes / 4, then the raw float32 rows).  The expected bytes are assembled by hand from that field list — this pins our writer / reader to the
    published layout; it is NOT a round trip through faiss (not installable here), so interoperability stays unverified."""
    import struct
    from kirag_amd.retriever import formats as I

    class Rows:                                                   # the three members write_faiss_flat_ip touches
        d = 4; ntotal = 3
        x = np.arange(12, dtype=np.float32).reshape(3, 4) / 8 - 0.5
        def reconstruct_n(self, s, n): return self.x[s:s + n]
    path = str(tmp_path / "index.faiss")
    I.write_faiss_flat_ip(Rows(), path)
    want = (b"IxFI" + struct.pack("<i", 4) + struct.pack("<q", 3) + struct.pack("<q", 1 << 20) * 2 + bytes([1]) + struct.pack("<i", 0)
            + struct.pack("<Q", 12) + Rows.x.tobytes())
    got = open(path, "rb").read()
    assert len(want) == 4 + (4 + 8 + 8 + 8 + 1 + 4) + 8 + 48 and got == want
    # the reader's header parse (device-free part): wrong fourcc / metric / payload count are rejected before any allocation
    for bad, msg in ((b"IxF2" + want[4:], "not a faiss IndexFlatIP"), (want[:33] + struct.pack("<i", 1) + want[37:], "metric_type"),
                     (want[:37] + struct.pack("<Q", 11) + want[45:], "payload")):
        with open(path, "wb") as f:
            f.write(bad)
        with pytest.raises(ValueError) as ei:
            I.read_faiss_flat_ip(path)
        assert msg in str(ei.value)
    assert I._SHARD_HEADER.size == 8 + 16 + 24 + 8 and I.shard_file_name(3, 8) == "index_shard_0003_of_0008.krshard"


def test_bulk_id_strings_equal_str_per_id():
    """``ids_to_str_rows`` (one ``kr_format_ids`` buffer, decoded and split once) == ``[[str(v) for v in row] for row in ids]`` — the list
    ``Indexer.search_knn`` returns per hit (index.py:49) — for every int64 incl. the extremes, negative ids and degenerate shapes."""
    from kirag_amd.retriever.index import ids_to_str_rows
    rng = np.random.default_rng(3)
    a = rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, size=(37, 11), dtype=np.int64)
    a[0, :4] = [0, -1, np.iinfo(np.int64).min, np.iinfo(np.int64).max]
    a[1] = 10 ** np.arange(11)
    out = ids_to_str_rows(a)
    assert out == [[str(v) for v in row] for row in a.tolist()] and all(type(v) is str for row in out for v in row)
    assert ids_to_str_rows(a[:, ::2]) == [[str(v) for v in row] for row in a[:, ::2].tolist()]          # non-contiguous view
    assert ids_to_str_rows(np.empty((3, 0), np.int64)) == [[], [], []] and ids_to_str_rows(np.empty((0, 5), np.int64)) == []
    # both implementations: the _fastids C extension (built next to the library when Python.h is present) and the kr_format_ids + split form behind it
    from kirag_amd.retriever import flat_index as F
    assert F._ids_to_str_rows_ascii(np.ascontiguousarray(a)) == out
    import glob
    if glob.glob(os.path.join(REPO, "kirag_amd", "_fastids*.so")):
        assert F._fastids is not None and F._fastids.ids_to_str_rows(memoryview(np.ascontiguousarray(a)).cast("B"), 37, 11) == out
        with pytest.raises(ValueError):
            F._fastids.ids_to_str_rows(memoryview(np.ascontiguousarray(a)).cast("B"), 37, 12)
    lib = _lib.load()
    import ctypes as C
    buf = C.create_string_buffer(8); n = C.c_int64(0)
    v = np.array([123456789012], np.int64)
    assert lib.kr_format_ids(v.ctypes.data, 1, b" ", C.addressof(buf), 8, C.byref(n)) == -22         # KR_EINVAL: fewer than 21 bytes left for an id
