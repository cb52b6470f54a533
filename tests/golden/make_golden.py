#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

Run:  PYTHONPATH=/root/reference python tests/golden/make_golden.py
The reference never travels: only the small input/output vectors written here are committed.
Weights are NOT stored — they are regenerated from the seeded recipe ``oracle.encoder_np.synth_weights``
and loaded into the reference's own classes with ``load_state_dict``.

What is captured:
  G1  E5Encoder / BGEEncoder forward on tiny configs (+ per-layer hidden states)   encoders.py:61-77,100-118
  G2  E5Encoder / BGEEncoder forward at the full e5-large shape (outputs only)
  G3  average_pool incl. an all-masked row -> NaN                                   encoders.py:56-58
  G4  BaseRetriever.compute_logits / score, four rank cases + ValueError            retrievers.py:71-91
  G5  BaseRetriever.encoder_embed 3-D path                                          retrievers.py:100-110
  G6  InBatchRetriever.forward single process (loss, scores)                        retrievers.py:133-150
  G7  E5Collator / BGECollator on a synthetic WordPiece vocab                       collators.py:59-89,132-156
  G8  DenseRetriever.calculate_*_embeddings batching                                retrievers.py:194-232
  G10 E5Encoder / BGEEncoder at the full shape: S = 256 / 512, B = 64, left padding, OUTLIER-channel weights (outputs only)

``retriever/retrievers.py`` imports ``retriever/index.py`` which imports ``faiss`` at module scope
(index.py:6,12-15).  faiss is not installed and cannot be: an EMPTY placeholder module (three attribute
names set to None) is put in ``sys.modules`` so that the import statement succeeds.  It emulates no faiss
behaviour and no Indexer golden is produced from it — the search half stays "parity unpinned".
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
REF = "/root/reference"
if REF not in sys.path:
    sys.path.insert(1, REF)

from oracle.encoder_np import synth_tokens, synth_weights, synth_weights_outlier  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

from transformers import BertConfig  # noqa: E402
from retriever.encoders import BGEEncoder, E5Encoder, average_pool  # noqa: E402


def build(cls, cfgd, seed, weights=None):
    cfg = BertConfig(vocab_size=cfgd["vocab"], hidden_size=cfgd["H"], num_hidden_layers=cfgd["L"],
                     num_attention_heads=cfgd["heads"], intermediate_size=cfgd["FF"],
                     max_position_embeddings=cfgd["max_pos"], type_vocab_size=2, layer_norm_eps=1e-12,
                     hidden_act="gelu")
    m = cls(cfg, add_pooling_layer=False).eval()
    w = weights if weights is not None else synth_weights(cfgd["H"], cfgd["L"], cfgd["FF"], cfgd["vocab"], cfgd["max_pos"], seed=seed)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "pooler" in k for k in missing), missing
    return m


TINY = {
    "t64": dict(H=64, L=2, heads=4, FF=256, vocab=1000, max_pos=512),
    "t128": dict(H=128, L=2, heads=2, FF=512, vocab=1000, max_pos=512),     # d_h = 64: runs on the HIP path
    "t256": dict(H=256, L=3, heads=4, FF=1024, vocab=2000, max_pos=512),    # d_h = 64: runs on the HIP path
}


def g1():
    out = {}
    for name, cfgd in TINY.items():
        for cls, tag in ((E5Encoder, "e5"), (BGEEncoder, "bge")):
            m = build(cls, cfgd, seed=11)
            for ci, (B, S, ragged) in enumerate([(1, 1, False), (3, 7, True), (8, 16, True), (5, 33, True),
                                                 (2, 128, False), (4, 128, True), (2, 512, True)]):
                ids, mask = synth_tokens(B, S, seed=100 + ci, ragged=ragged, vocab_lo=5, vocab_hi=cfgd["vocab"], min_len=1)
                ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
                with torch.no_grad():
                    o = m(ti, tm)
                key = f"{name}.{tag}.c{ci}"
                out[key + ".ids"] = ids.astype(np.int32)
                out[key + ".mask"] = mask.astype(np.int8)
                out[key + ".out"] = o.numpy()
                if tag == "e5" and ci in (1, 3):
                    from transformers import BertModel
                    with torch.no_grad():
                        hs = BertModel.forward(m, input_ids=ti, attention_mask=tm, output_hidden_states=True,
                                               return_dict=True).hidden_states
                    out[key + ".hidden"] = np.stack([h.numpy() for h in hs])
    out["cfg_names"] = np.array(list(TINY.keys()))
    for name, cfgd in TINY.items():
        out[f"cfg.{name}"] = np.array([cfgd[k] for k in ("H", "L", "heads", "FF", "vocab", "max_pos")])
    out["weight_seed"] = np.array(11)
    np.savez_compressed(os.path.join(OUT, "g1_encoder_tiny.npz"), **out)
    print("G1", len(out), "arrays")


def g2():
    cfgd = dict(H=1024, L=24, heads=16, FF=4096, vocab=30522, max_pos=512)
    out = {"cfg": np.array([cfgd[k] for k in ("H", "L", "heads", "FF", "vocab", "max_pos")]), "weight_seed": np.array(0)}
    for cls, tag in ((E5Encoder, "e5"), (BGEEncoder, "bge")):
        m = build(cls, cfgd, seed=0)
        cases = [(8, 128, False, 1), (8, 128, True, 1), (4, 32, True, 2)] if tag == "e5" else [(8, 128, True, 1)]
        for ci, (B, S, ragged, seed) in enumerate(cases):
            ids, mask = synth_tokens(B, S, seed=seed, ragged=ragged)
            with torch.no_grad():
                o = m(torch.from_numpy(ids), torch.from_numpy(mask))
            key = f"{tag}.c{ci}"
            out[key + ".ids"] = ids.astype(np.int32); out[key + ".mask"] = mask.astype(np.int8); out[key + ".out"] = o.numpy()
        del m
    np.savez_compressed(os.path.join(OUT, "g2_encoder_large.npz"), **out)
    print("G2 done")


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import g10_spec  # noqa: E402  (tables of weights / cases / token layouts shared with tests/test_gpu_encoder.py)


def g10():
    """Full e5-large / bge-large shape beyond G2: S = 256 / 512, a 64-sequence batch, left padding, and weights with OUTLIER channels
    (oracle.encoder_np.synth_weights_outlier: LayerNorm gamma 8-16x / 30-60x in six hidden channels, matching embedding columns, large biases)
    — the regime where 16-bit operands / a 16-bit residual stream lose the most.  Outputs + per-case hidden-state magnitudes only."""
    cfgd = dict(g10_spec.CFG)
    out = {"cfg": np.array([cfgd[k] for k in ("H", "L", "heads", "FF", "vocab", "max_pos")])}
    only = [w for w in os.environ.get("G10_ONLY", "").split(",") if w]      # regenerate / add only these weight sets, keep the rest of the file
    path = os.path.join(OUT, "g10_encoder_large_ext.npz")
    if only and os.path.exists(path):
        old = np.load(path)
        out.update({k: old[k] for k in old.files if k.split(".")[0] not in only})
    for wname in g10_spec.WEIGHTS:
        if only and wname not in only:
            continue
        w = g10_spec.weights(wname)
        for cls, tag in ((E5Encoder, "e5"), (BGEEncoder, "bge")):
            m = build(cls, cfgd, None, weights=w)
            for ci, (B, S, layout, seed) in enumerate(g10_spec.CASES[wname][tag]):
                ids, mask = g10_spec.tokens(B, S, layout, seed)
                from transformers import BertModel
                with torch.no_grad():
                    o = m(torch.from_numpy(ids), torch.from_numpy(mask))
                    hs = BertModel.forward(m, input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                                           output_hidden_states=True, return_dict=True).hidden_states if ci == 0 else None
                key = f"{wname}.{tag}.c{ci}"
                out[key + ".out"] = o.numpy()
                if hs is not None:          # statistics of the residual stream of the first sequence: documents how hard the recipe is
                    out[key + ".absmax"] = np.array([float(h[0].abs().max()) for h in hs], np.float32)
                    out[key + ".absmed"] = np.array([float(h[0].abs().median()) for h in hs], np.float32)
                print("G10", key, (B, S, layout), "done", flush=True)
            del m
        del w
    np.savez_compressed(os.path.join(OUT, "g10_encoder_large_ext.npz"), **out)
    print("G10 done")


def g3():
    rng = np.random.Generator(np.random.PCG64(3))
    lh = rng.standard_normal((4, 6, 16)).astype(np.float32)
    mask = np.array([[1, 1, 1, 1, 1, 1], [1, 1, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 1, 1, 1, 1]], np.int64)
    o = average_pool(torch.from_numpy(lh), torch.from_numpy(mask)).numpy()
    assert np.isnan(o[2]).all()
    np.savez_compressed(os.path.join(OUT, "g3_average_pool.npz"), lh=lh, mask=mask, out=o)
    print("G3 done")


def _import_retrievers():
    if "faiss" not in sys.modules:
        ph = types.ModuleType("faiss")  # import-time placeholder only: see module docstring
        ph.IndexFlatIP = None; ph.IndexFlatL2 = None; ph.IndexPQ = None
        sys.modules["faiss"] = ph
    import retriever.retrievers as rr
    return rr


def _save_tiny_model(cls, cfgd, seed, path):
    m = build(cls, cfgd, seed)
    m.save_pretrained(path)


def _synthetic_vocab():
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "query", "passage", ":", ",", ".", "title", "text",
             "represent", "this", "sentence", "for", "searching", "relevant", "passages", "the", "of", "and", "in",
             "who", "what", "when", "where", "is", "was", "born", "city", "film", "director", "river", "capital",
             "##s", "##ed", "##ing", "a", "b", "c", "d", "e", "f", "g", "h", "i", "j", "k", "l", "m", "n", "o", "p",
             "q", "r", "s", "t", "u", "v", "w", "x", "y", "z", "##a", "##b", "##c", "##d", "##e", "##f", "##g", "##h",
             "##i", "##j", "##k", "##l", "##m", "##n", "##o", "##p", "##q", "##r", "##t", "##u", "##v", "##w",
             "##x", "##y", "##z", "0", "1", "2", "3", "4", "5", "6", "7", "8", "9", "?", "paris", "france", "london"]
    return words


def g4_to_g8():
    rr = _import_retrievers()
    from dataset.collators import BGECollator, E5Collator
    from kirag_amd.bench_support import wordpiece_tokenizer
    cfgd = dict(TINY["t128"]); vocab = _synthetic_vocab(); cfgd["vocab"] = len(vocab)
    out = {"vocab": np.array(vocab), "cfg": np.array([cfgd[k] for k in ("H", "L", "heads", "FF", "vocab", "max_pos")]),
           "weight_seed": np.array(21)}
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "vocab.txt"), "w") as f:
            f.write("\n".join(vocab) + "\n")
        tok = wordpiece_tokenizer(os.path.join(td, "vocab.txt"))
        _save_tiny_model(E5Encoder, cfgd, 21, td)
        ret = rr.InBatchRetriever("E5Retriever", td, temperature=0.01).eval()
        # G4
        rng = np.random.Generator(np.random.PCG64(4))
        q1 = rng.standard_normal(16).astype(np.float32); d1 = rng.standard_normal(16).astype(np.float32)
        q2 = rng.standard_normal((3, 16)).astype(np.float32); d2 = rng.standard_normal((5, 16)).astype(np.float32)
        d3 = rng.standard_normal((3, 4, 16)).astype(np.float32)
        T = torch.from_numpy
        out.update({"g4.q1": q1, "g4.d1": d1, "g4.q2": q2, "g4.d2": d2, "g4.d3": d3,
                    "g4.l11": ret.compute_logits(T(q1), T(d1)).numpy(), "g4.l12": ret.compute_logits(T(q1), T(d2)).numpy(),
                    "g4.l23": ret.compute_logits(T(q2), T(d3)).numpy(), "g4.l22": ret.compute_logits(T(q2), T(d2)).numpy(),
                    "g4.s22_t001": ret.score(T(q2), T(d2)).numpy()})
        ret.temperature = "sqrt"
        out["g4.s22_sqrt"] = ret.score(T(q2), T(d2)).numpy()
        ret.temperature = 0.01
        try:
            ret.compute_logits(T(d3), T(d3)); raise SystemExit("expected ValueError")
        except ValueError as e:
            out["g4.err"] = np.array(str(e))
        # G7 collators
        e5c = E5Collator(tokenizer=tok, query_maxlength=16, doc_maxlength=24)
        bgc = BGECollator(tokenizer=tok, query_maxlength=24, doc_maxlength=24)
        queries = ["who was born in paris?", "what is the capital of france", "river"]
        docs = ["title:  paris, text:  paris is the capital city of france.", "title:  london, text:  a city",
                "title:  x, text:  " + "the river " * 20]
        for nm, col in (("e5", e5c), ("bge", bgc)):
            qa = col.encode_query(queries); da = col.encode_doc(docs); qs = col.encode_query(queries, max_length=8)
            out[f"g7.{nm}.q.ids"] = qa["input_ids"].numpy(); out[f"g7.{nm}.q.mask"] = qa["attention_mask"].numpy()
            out[f"g7.{nm}.d.ids"] = da["input_ids"].numpy(); out[f"g7.{nm}.d.mask"] = da["attention_mask"].numpy()
            out[f"g7.{nm}.q8.ids"] = qs["input_ids"].numpy()
        out["g7.queries"] = np.array(queries); out["g7.docs"] = np.array(docs)
        # G5 3-D encoder_embed
        ids, mask = synth_tokens(6, 12, seed=5, ragged=True, vocab_lo=5, vocab_hi=len(vocab), min_len=2)
        a3 = {"input_ids": T(ids).reshape(2, 3, 12), "attention_mask": T(mask).reshape(2, 3, 12)}
        with torch.no_grad():
            out["g5.out"] = ret.doc(a3).numpy()
        out["g5.ids"] = ids.reshape(2, 3, 12); out["g5.mask"] = mask.reshape(2, 3, 12)
        # G6 InBatchRetriever.forward
        qa = e5c.encode_query(queries); da = e5c.encode_doc(docs)
        labels = torch.tensor([0, 1, 2])
        with torch.no_grad():
            loss, scores, gq, gd = ret(qa, da, labels)
        out["g6.loss"] = loss.numpy(); out["g6.scores"] = scores.numpy(); out["g6.q"] = gq.numpy(); out["g6.d"] = gd.numpy()
        # G8 DenseRetriever embedding batching (batch_size=2 over 3 items -> per-batch padding)
        dr = rr.DenseRetriever(retriever=ret, collator=e5c, indexer=None, corpus=None, batch_size=2)
        out["g8.qemb"] = dr.calculate_query_embeddings(queries).numpy()
        out["g8.demb"] = dr.calculate_document_embeddings(docs, max_length=16).numpy()
    np.savez_compressed(os.path.join(OUT, "g4_g8_retriever.npz"), **out)
    print("G4-G8 done")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g48"]
    if "g1" in which: g1()
    if "g3" in which: g3()
    if "g48" in which: g4_to_g8()
    if "g2" in which: g2()
    if "g10" in which: g10()
