"""Host side of the corpus-encode feed: ragged token frames from tokenizer processes into a pinned ring
(``kirag_amd/feed.py``), the worker protocol (``kirag_amd/tokenize_worker.py``), and ``cal_doc_embeddings``' file contract
(``compute_corpus_embeddings.py:101-120``) through every feed variant.  No GPU: the frames are checked against the collator's own output
(``dataset/collators.py:59-81,143-145`` semantics), the encoder is a deterministic stand-in."""
import os
import pickle
import struct
import threading
import time
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from kirag_amd import compute_corpus_embeddings as CC
from kirag_amd import feed
from kirag_amd.bench_support import wordpiece_tokenizer
from kirag_amd.collators import E5Collator

WORDS = ["hello", "world", "foo", "bar", "bars", "passage", "title", "text"]


@pytest.fixture(scope="module")
def collator(tmp_path_factory):
    d = tmp_path_factory.mktemp("vocab")
    with open(d / "vocab.txt", "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "passage", ":", ",", "hello", "world", "foo", "bar", "##s", "title", "text"]) + "\n")
    return E5Collator(tokenizer=wordpiece_tokenizer(str(d / "vocab.txt")), query_maxlength=16, doc_maxlength=12)


def texts_of(n):
    return [" ".join(np.random.default_rng(i).choice(WORDS, 1 + i % 14)) for i in range(n)]


def test_tokens_of_ragged_padded_and_frame_round_trip():
    ids = torch.tensor([[2, 8, 9, 3, 0, 0], [2, 8, 3, 0, 0, 0], [2, 3, 7, 7, 7, 7]])
    mask = torch.tensor([[1, 1, 1, 1, 0, 0], [1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1]])
    t = feed.tokens_of({"input_ids": ids, "attention_mask": mask})
    assert t.kind == feed.KIND_RAGGED and (t.n, t.S, t.T) == (3, 6, 13)
    assert t.lens.tolist() == [4, 3, 6] and t.ids.tolist() == [2, 8, 9, 3, 2, 8, 3, 2, 3, 7, 7, 7, 7] and t.ids.dtype == np.int32
    back = feed.repad(t.ids, t.lens, t.S, 0)
    assert torch.equal(back["input_ids"], ids * mask) and torch.equal(back["attention_mask"], mask)
    # frame bytes: header + lens + ids, nothing else; 4-16x smaller than the padded int64 pair
    blob = b"".join(feed.pack_frame(t))
    magic, kind, n, S, T, aux = feed._HEAD.unpack(blob[:feed._HEAD.size])
    assert (magic, kind, n, S, T, aux) == (feed.FRAME_MAGIC, feed.KIND_RAGGED, 3, 6, 13, 0)
    assert len(blob) == feed._HEAD.size + 4 * 3 + 4 * 13 < ids.numel() * 16
    assert np.frombuffer(blob, np.int32, 3, feed._HEAD.size).tolist() == [4, 3, 6]
    # a mask that is not 1^len 0^(S-len) (left padding, a hole): travels padded, nothing is re-ordered
    for m in (torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1]]), torch.tensor([[1, 0, 1, 1, 0, 0], [1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1]])):
        p = feed.tokens_of({"input_ids": ids, "attention_mask": m})
        assert p.kind == feed.KIND_PADDED and p.T == 18 and p.ids.tolist() == ids.reshape(-1).tolist() and p.mask.tolist() == m.reshape(-1).tolist()
    # all-masked rows are legal (the reference yields NaN for them): length 0
    z = feed.tokens_of({"input_ids": ids, "attention_mask": torch.zeros_like(mask)})
    assert z.kind == feed.KIND_RAGGED and z.T == 0 and z.lens.tolist() == [0, 0, 0] and feed.attended_range(z) is None
    with pytest.raises(ValueError):
        feed.tokens_of({"input_ids": ids, "attention_mask": mask[:, :3]})


def test_frame_validation_looks_at_attended_positions_only():
    ids = torch.tensor([[2, 8, 3, 500], [2, 3, 500, 500]]); mask = torch.tensor([[1, 1, 1, 0], [1, 1, 0, 0]])
    t = feed.tokens_of({"input_ids": ids, "attention_mask": mask})
    assert feed.attended_range(t) == (2, 8)
    assert feed._HEAD.unpack(feed.pack_frame(t, vocab=100)[0])[1] == feed.KIND_RAGGED          # 500 sits at masked positions only
    bad = feed.tokens_of({"input_ids": torch.tensor([[2, 150, 3, 0], [2, -4, 0, 0]]), "attention_mask": mask})
    parts = feed.pack_frame(bad, vocab=100)
    magic, kind, n, S, T, aux = feed._HEAD.unpack(parts[0])
    assert kind == feed.KIND_BAD_ID and len(parts) == 1
    assert struct.unpack("<ii", struct.pack("<II", (aux >> 32) & 0xffffffff, aux & 0xffffffff)) == (-4, 150)
    # an id beyond int32 must not wrap into the vocabulary when the frame narrows to int32
    wide = feed.tokens_of({"input_ids": torch.tensor([[2, 2**32 + 5, 3, 0], [2, 3, 0, 0]]), "attention_mask": mask})
    assert feed.attended_range(wide) == (2, 2**31 - 1) and feed._HEAD.unpack(feed.pack_frame(wide, vocab=100)[0])[1] == feed.KIND_BAD_ID
    # left-padded frames are validated through their mask
    lp = feed.tokens_of({"input_ids": torch.tensor([[900, 2, 3], [2, 8, 3]]), "attention_mask": torch.tensor([[0, 1, 1], [1, 1, 1]])})
    assert lp.kind == feed.KIND_PADDED and feed.attended_range(lp) == (2, 8)


@pytest.mark.parametrize("workers", [0, 3])
def test_token_feed_frames_arrive_in_order_and_equal_the_collator(collator, workers):
    """Frames of a 23-batch corpus through a ring of few slots (every slot is reused several times): consumed in order, each equal to the in-process
    collator output; the slot of a frame is not overwritten before its release event has been waited for."""
    texts = texts_of(227)

    def mk(s):
        return texts[s:s + 10], [str(i) for i in range(s, min(s + 10, 227))]
    items = list(range(0, 227, 10))
    tf = feed.TokenFeed(mk, collator, items, workers, 1, 10, 12, vocab=15, local=False)     # workers > 0: the worker processes alone (the local thread would finish
    assert tf.R <= workers + 4                                                              # these 23 tiny batches before a worker has imported torch)

    class SlowEvent:                                # stands for the upload's completion: the feed must wait for it before handing the slot on
        def __init__(self): self.done = False
        def synchronize(self):
            time.sleep(0.01); self.done = True
    seen, events = [], {}
    for frame in tf:
        s = items[frame.index]
        ref = collator.encode_doc(texts[s:s + 10])
        got = frame.inputs(pad_id=0)
        assert frame.kind == feed.KIND_RAGGED and frame.ids.dtype == torch.int32 and frame.lens.dtype == torch.int32
        assert torch.equal(got["input_ids"], ref["input_ids"]) and torch.equal(got["attention_mask"], ref["attention_mask"])
        assert frame.S == ref["input_ids"].shape[1] and frame.T == int(ref["attention_mask"].sum()) and frame.doc_ids == mk(s)[1]
        prev = events.get(id(frame.slot))
        assert prev is None or prev.done, "a slot was reused before the event of its previous frame completed"
        ev = SlowEvent(); events[id(frame.slot)] = ev
        tf.release(frame, ev)
        seen.append(frame.index)
    assert seen == list(range(len(items))) and tf.made_by == ({"local": 23, "workers": 0} if workers == 0 else {"local": 0, "workers": 23})
    assert not tf.procs and all(not t.is_alive() or t.join(2) is None for t in tf.threads)


@pytest.mark.parametrize("workers", [0, 2])
def test_token_feed_reports_a_bad_batch_at_its_turn_and_worker_failures(collator, workers):
    texts = texts_of(60)

    def mk(s):
        return texts[s:s + 10], list(range(s, s + 10))
    # vocab = 12: "title" / "text" (ids 13, 14) only occur in batch 3 -> batches 0..2 come through, batch 3 raises
    clean = [" ".join(w for w in t.split() if w not in ("title", "text")) or "foo" for t in texts]
    clean[34] += " title"

    def mk2(s):
        return clean[s:s + 10], list(range(s, s + 10))
    got = []
    with pytest.raises(ValueError, match=r"starting at passage id 30 contain a token id outside \[0, 13\) \(min 2, max 13\)"):
        for f in feed.TokenFeed(mk2, collator, range(0, 60, 10), workers, 2, 10, 12, vocab=13, local=False):
            got.append(f.index)
    assert got == [0, 1, 2]
    # an exception inside the tokenizer (an empty batch: the collator raises) surfaces at the consumer, the feed shuts down
    with pytest.raises((RuntimeError, ValueError), match="text_list is None or an empty"):
        list(feed.TokenFeed(lambda s: ([], [s]), collator, range(3), workers, 1, 10, 12, local=False))
    # abandoning the iterator stops workers and threads
    n0 = threading.active_count()
    tf = feed.TokenFeed(mk, collator, range(0, 60, 10), workers, 1, 10, 12)
    it = iter(tf); next(it); it.close()
    time.sleep(0.3)
    assert not tf.procs and threading.active_count() <= n0 + 1


class _Bag(torch.nn.Module):
    """Deterministic stand-in encoder: embedding = normalised bag of attended token ids; ``doc_packed`` is defined but must not be called off the GPU."""
    def __init__(self, d=16):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(64, d, generator=torch.Generator().manual_seed(0)))
        self.encoder = SimpleNamespace(config=SimpleNamespace(vocab_size=64))
    def doc(self, a, **k):
        m = a["attention_mask"].unsqueeze(-1).float()
        return torch.nn.functional.normalize((self.w[a["input_ids"] % 64] * m).sum(1), dim=-1)
    def doc_packed(self, *a, **k):
        raise AssertionError("the packed forward is the GPU path")


def test_shard_files_are_identical_through_every_feed(tmp_path, collator):
    """cal_doc_embeddings' files (compute_corpus_embeddings.py:101-120) for one corpus: in-process collate (the rounds 1-5 loop) and tokenizer processes with
    ragged frames at two ring depths give the same files — same names, the id lists equal as bytes, the tensors equal bit for bit with the same dtype, shape,
    strides and an exactly-sized storage.  (The tensor pickles themselves are never equal as bytes, not even between two runs of the same code: torch's zip
    container carries a per-save ``serialization_id``.)"""
    n = 131
    texts = texts_of(n)

    class Corpus:
        index_to_passage_id = {i: str(3 * i + 7) for i in range(n)}
        def __len__(self): return n
        def __getitem__(self, i): return {"index": i, "passage": texts[i]}

    def run(name, **kw):
        args = SimpleNamespace(local_rank=-1, save_dir=str(tmp_path), name=name, index_folder="f", per_gpu_batch_size=8, num_passage_per_index_file=50,
                               encode_batch_size=16, **kw)
        assert CC.cal_doc_embeddings(args, _Bag(), Corpus(), collator, device=torch.device("cpu")) == (0, n)
        folder = os.path.join(str(tmp_path), name, "f")
        out = {}
        for f in sorted(os.listdir(folder)):
            raw = open(os.path.join(folder, f), "rb").read()
            out[f] = raw if f.startswith("passage_id_list_") else pickle.loads(raw)
        return out

    def same(a, b):
        assert sorted(a) == sorted(b)
        for f in a:
            if f.startswith("passage_id_list_"):
                assert a[f] == b[f], f
            else:
                x, y = a[f], b[f]
                assert x.dtype == y.dtype == torch.float32 and x.shape == y.shape and x.stride() == y.stride(), f
                assert x.untyped_storage().nbytes() == y.untyped_storage().nbytes() == x.numel() * 4, f
                assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f
    base = run("inproc", tokenizer_workers=0)
    assert sorted(base) == ["corpus_embeddings_0_49.pkl", "corpus_embeddings_100_130.pkl", "corpus_embeddings_50_99.pkl",
                            "passage_id_list_0_49.pkl", "passage_id_list_100_130.pkl", "passage_id_list_50_99.pkl"]
    same(run("procs", tokenizer_workers=3, prefetch_batches=1), base)
    same(run("buffered", tokenizer_workers=0, buffered_shard_files=True), base)          # pickle.dump of a staged tensor (rounds 1-5) == the streamed file
    assert not [f for f in os.listdir(os.path.join(str(tmp_path), "inproc", "f")) if f.endswith(".tmp")]
    e = base["corpus_embeddings_100_130.pkl"]
    assert isinstance(e, torch.Tensor) and tuple(e.shape) == (31, 16)
    assert pickle.loads(base["passage_id_list_100_130.pkl"]) == [str(3 * i + 7) for i in range(100, 131)]
    ref = _Bag().doc(collator.encode_doc(texts[100:131])).detach()
    torch.testing.assert_close(e, ref)


def test_local_thread_and_workers_share_one_stream_and_a_worker_that_cannot_start_is_not_fatal(collator, caplog):
    """Dynamic hand-out: the local tokenizer thread carries the stream while worker processes start and both kinds produce frames of one ordered stream; a
    collator that cannot be pickled (or a worker that dies on start) costs the speed-up, not the run."""
    texts = texts_of(400)

    def mk(s):
        time.sleep(0.05)                       # slow dataset access: keeps the stream alive long enough for the workers to join in
        return texts[s:s + 4], list(range(s, s + 4))
    tf = feed.TokenFeed(mk, collator, range(0, 400, 4), 2, 2, 4, 12)
    for k, frame in enumerate(tf):
        assert frame.index == k
        ref = collator.encode_doc(texts[4 * k:4 * k + 4])
        assert torch.equal(frame.inputs()["input_ids"], ref["input_ids"])
    assert tf.made_by["local"] > 0 and tf.made_by["local"] + tf.made_by["workers"] == 100
    # (whether a worker got a batch depends on how fast this machine imports torch; the pure-worker path is covered above)

    class Unpicklable(E5Collator):
        def __reduce__(self):
            raise TypeError("no pickle for you")
    col2 = Unpicklable(tokenizer=collator.tokenizer, query_maxlength=16, doc_maxlength=12)
    with caplog.at_level("WARNING"):
        out = [f.index for f in feed.TokenFeed(lambda s: (texts[s:s + 4], [s]), col2, range(0, 40, 4), 2, 1, 4, 12)]
    assert out == list(range(10)) and "tokenising in-process only" in caplog.text
    with pytest.raises(TypeError):
        list(feed.TokenFeed(lambda s: (texts[s:s + 4], [s]), col2, range(0, 40, 4), 2, 1, 4, 12, local=False))


def test_default_tokenizer_workers_and_writer_errors(tmp_path):
    assert CC.default_tokenizer_workers(10**7, on_gpu=False) == 0 and CC.default_tokenizer_workers(199_999, on_gpu=True) == 0
    w = CC.default_tokenizer_workers(10**7, on_gpu=True)
    assert 0 <= w <= 4 and w <= max(1, len(os.sched_getaffinity(0)) // 4) and 1 <= CC.effective_cpus() <= len(os.sched_getaffinity(0))
    assert CC.setup_parser([]).tokenizer_workers == -1
    # a failure on the writer thread (here: the folder vanished) reaches the encode thread
    wr = CC._ShardWriter(str(tmp_path / "missing" / "dir"), 0, 8, 4, 4, on_gpu=False)
    host = wr.host_buffer(8); host.zero_()
    wr.put(None, host, 4, ["a", "b", "c", "d"])
    with pytest.raises(FileNotFoundError):
        wr.close()


def test_streaming_tensor_pickle_is_a_tensor_pickle(tmp_path):
    """kirag_amd/tensor_pickle.py: the streamed ``corpus_embeddings_*.pkl`` is what the reference's reader opens (faiss_index_corpus.py:38 ``pickle.load`` -> fp32
    ``torch.Tensor [n, hidden]``): loads equal to ``pickle.dump(tensor)``'s result (values bit for bit, dtype, shape, strides, exactly-sized storage, no grad), the
    storage container head is torch's own byte for byte (selftest), row blocks of any size may be appended, and a stream that is not completed leaves nothing."""
    from kirag_amd import tensor_pickle as TP
    assert TP.selftest() and TP.available()
    rng = np.random.default_rng(0)
    for rows, d in ((1, 16), (1000, 1024), (257, 384)):
        x = rng.standard_normal((rows, d)).astype(np.float32)
        x[0, 0] = np.nan; x[-1, -1] = -0.0
        p = str(tmp_path / f"s_{rows}_{d}.pkl")
        w = TP.StreamingTensorPickle(p, rows, d)
        cuts = sorted(set([0, rows] + rng.integers(0, rows + 1, 5).tolist()))
        for a, b in zip(cuts[:-1], cuts[1:]):
            w.append(x[a:b])
        w.close()
        assert not os.path.exists(p + ".tmp")
        got = pickle.load(open(p, "rb"))
        ref = pickle.loads(pickle.dumps(torch.from_numpy(x.copy())))
        assert type(got) is torch.Tensor and got.dtype == ref.dtype and got.shape == ref.shape and got.stride() == ref.stride() and not got.requires_grad
        assert got.untyped_storage().nbytes() == rows * d * 4 and torch.equal(got.view(torch.int32), ref.view(torch.int32))
        assert abs(os.path.getsize(p) - len(pickle.dumps(torch.from_numpy(x)))) < 64           # the same container around the same bytes
    w = TP.StreamingTensorPickle(str(tmp_path / "short.pkl"), 10, 4)
    w.append(np.zeros((3, 4), np.float32))
    with pytest.raises(ValueError):
        w.append(np.zeros((8, 4), np.float32))                  # more rows than announced
    with pytest.raises(ValueError):
        w.append(np.zeros((2, 4), np.float64))
    with pytest.raises(RuntimeError, match="3 of the 10"):
        w.close()
    assert not os.path.exists(str(tmp_path / "short.pkl")) and not os.path.exists(str(tmp_path / "short.pkl.tmp"))


def test_collator_fast_path_equals_the_reference_call_and_switches_itself_off_on_a_mismatch(collator, caplog):
    """``RetrieverCollator.encode``'s fast path (the Rust tokenizer called directly, padded with numpy) against the reference's call
    ``tokenizer(texts, max_length, padding, truncation=True, return_tensors="pt")`` (dataset/collators.py:76-80): the same int64 tensors for ragged batches,
    truncated texts, a single string, ``padding="max_length"`` and the ``max_length`` override; the ragged form equals the padded one stripped; a tokenizer for
    which the two differ (here: one that pads on the left) keeps the reference call."""
    texts = texts_of(64) + ["hello " * 40, "", "bars"]
    for col, maxlen in ((collator, None), (collator, 7), (E5Collator(tokenizer=collator.tokenizer, query_maxlength=9, doc_maxlength=9, doc_padding="max_length"), None)):
        kw = {} if maxlen is None else {"max_length": maxlen}
        got = col.encode_doc(texts, **kw)
        ml = maxlen or col.doc_maxlength
        ref = col._encode_reference([col.doc_prefix + t for t in texts], ml, col.doc_padding)
        assert torch.equal(got["input_ids"], ref["input_ids"]) and torch.equal(got["attention_mask"], ref["attention_mask"]) and got["input_ids"].dtype == torch.int64
        assert (ml, col.doc_padding) in col._fast_checked and not getattr(col, "_fast_off", False)
        rows, S = col.encode_doc_ragged(texts, **kw)
        t = feed.tokens_from_rows(rows, S); r = feed.tokens_of(ref)
        assert (t.kind, t.n, t.S, t.T) == (r.kind, r.n, r.S, r.T) and np.array_equal(t.ids, r.ids) and np.array_equal(t.lens, r.lens)
    q = collator.encode_query(["which bar ?"]); rq = collator._encode_reference(["query: which bar ?"], 16, "max_sequence")
    assert torch.equal(q["input_ids"], rq["input_ids"]) and torch.equal(q["attention_mask"], rq["attention_mask"])
    # a left-padding tokenizer: the fast path is not taken at all; a backend whose output differs: compared once, switched off, reference results from then on
    import copy
    left = copy.deepcopy(collator.tokenizer); left.padding_side = "left"
    cl = E5Collator(tokenizer=left, query_maxlength=16, doc_maxlength=12)
    out = cl.encode_doc(texts[:5])
    assert out["attention_mask"][0, 0] == 0 or out["attention_mask"].all() and cl.encode_doc_ragged(texts[:5]) is None and not getattr(cl, "_fast_checked", None)

    class Odd(E5Collator):
        def _pad_rows(self, rows, maxlength, padding):                      # stands for "this backend tokenises differently": shifted ids
            out = super()._pad_rows(rows, maxlength, padding)
            out["input_ids"] = out["input_ids"] + 1
            return out
    odd = Odd(tokenizer=collator.tokenizer, query_maxlength=16, doc_maxlength=12)
    with caplog.at_level("WARNING"):
        o = odd.encode_doc(texts[:5])
    r = collator._encode_reference(["passage: " + t for t in texts[:5]], 12, "max_sequence")
    assert torch.equal(o["input_ids"], r["input_ids"]) and odd._fast_off and "switched off" in caplog.text and odd.encode_doc_ragged(texts[:5]) is None


def test_a_tokenizer_process_that_dies_mid_stream_fails_the_run_loudly(collator):
    """A worker killed while it owes a batch: the consumer gets a RuntimeError at that batch's turn (never a silently shorter or re-ordered stream), and the feed shuts down."""
    texts = texts_of(200)

    def mk(s):
        time.sleep(0.02)
        return texts[s:s + 4], list(range(s, s + 4))
    tf = feed.TokenFeed(mk, collator, range(0, 200, 4), 1, 1, 4, 12, local=False)
    got = []
    with pytest.raises(RuntimeError, match="exited unexpectedly|Broken pipe|tokenizer worker"):
        for frame in tf:
            got.append(frame.index)
            if frame.index == 3:
                tf.procs[0].kill()
    assert got[:4] == [0, 1, 2, 3] and len(got) < 50 and got == list(range(len(got)))
    assert not tf.procs
