#!/usr/bin/env python3
"""One KiRAG retrieval hop at the REFERENCE'S OWN SURFACE (knowledge_graph/models.py:1645: ``self.retriever(queries, topk)`` = DenseRetriever.forward ->
batch_retrieve, retrievers.py:250-291): strings in, lists of {"id", "score"} out — tokenizer, collator, E5Encoder module forward, D2H, Indexer.search_knn, result
parsing — over a resident N-row index, next to the C-ABI pieces (HipBertForward.forward + FlatIPIndex.search) on the same tokens.  The difference is host work
of the surface.  Usage: python tools/hop_surface.py [total_rows] [--profile]"""
import os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import torch.nn as nn
from transformers import BertConfig
from kirag_amd.bench_support import wordpiece_tokenizer
from kirag_amd.collators import E5Collator
from kirag_amd.retriever.encoders import E5Encoder
from kirag_amd.retriever.index import Indexer
from kirag_amd.retriever.retrievers import BaseRetriever, DenseRetriever

total = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5_000_000
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
td = tempfile.mkdtemp()
from kirag_amd.bench_support import synthetic_text_corpus
vocab_file, _texts = synthetic_text_corpus(8, td, seed=5)
words = [l.strip() for l in open(vocab_file) if l.strip().isalpha()]
tok = wordpiece_tokenizer(vocab_file)
cfg = BertConfig(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, max_position_embeddings=512)
enc = E5Encoder(cfg, add_pooling_layer=False).to(dev).eval()

class Ret(BaseRetriever):
    def __init__(self, encoder):
        nn.Module.__init__(self)
        self.encoder = encoder
        self.norm_query = self.norm_doc = False
        self.temperature, self.local_rank, self.world_size = 1.0, -1, 1

ix = Indexer(1024)
g = torch.Generator(device=dev); g.manual_seed(3)
ix.index.reserve(total)
for s0 in range(0, total, 250_000):
    m = min(250_000, total - s0)
    ix.index.add(torch.nn.functional.normalize(torch.randn(m, 1024, generator=g, device=dev), dim=1))
ix.index_id_to_db_id = np.arange(total, dtype=np.int64) * 3 + 1_000_000
col = E5Collator(tokenizer=tok, query_maxlength=256, doc_maxlength=128)
dr = DenseRetriever(retriever=Ret(enc), collator=col, indexer=ix, corpus=None, batch_size=4)
wa = np.array(words)
question = "which " + " ".join(rng.choice(wa, 12)) + " ?"
chain = ". ".join("<" + " ".join(rng.choice(wa, 9)) + ">" for _ in range(16))
query = "{}\nknowledge triples: {}.".format(question, chain)
ntok = int(col.encode_query([query], max_length=256)["attention_mask"].sum())

def hop_surface(): return dr([query], 10)
a = col.encode_query([query], max_length=256)
ids, mask = a["input_ids"].to(dev), a["attention_mask"].to(dev)
def hop_abi():
    qv = enc._hip.forward(ids, mask, 0)
    return ix.index.search(qv, 10)

def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
hop_surface()
t_surface, t_abi = timed(hop_surface), timed(hop_abi)
t_tok = timed(lambda: col.encode_query([query], max_length=256))
t_emb = timed(lambda: dr.calculate_query_embeddings(queries=[query], max_length=256))
qe = dr.calculate_query_embeddings(queries=[query], max_length=256).numpy()
t_knn = timed(lambda: ix.search_knn(qe, 10, verbose=False))
# the collator's fast path on / off, interleaved on this box
ab = {True: [], False: []}
for rep in range(6):
    for fast in (True, False):
        col._fast_off = not fast
        ab[fast].append((timed(hop_surface, 20), timed(lambda: col.encode_query([query], max_length=256), 20)))
col._fast_off = False
for fast in (True, False):
    hs, ts = zip(*ab[fast])
    print(f"  collator fast path {'on ' if fast else 'off'}: hop {np.median(hs):.2f} ms (runs {' '.join('%.2f' % h for h in hs)}), tokenizer + collator {np.median(ts):.3f} ms")
print(f"one hop, nq = 1, query of {ntok} tokens, top-10 over {total} rows:")
print(f"  reference surface  DenseRetriever([query], 10)                       {t_surface:.2f} ms")
print(f"  C ABI              HipBertForward.forward + FlatIPIndex.search         {t_abi:.2f} ms")
print(f"  pieces of the surface: tokenizer + collator {t_tok:.2f} ms | calculate_query_embeddings (tokenize, H2D, module forward, D2H) {t_emb:.2f} ms | Indexer.search_knn {t_knn:.2f} ms")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): hop_surface()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)

# stage by stage (perf_counter around the pieces batch_retrieve runs, 200 hops; the GPU work is asynchronous until search_knn waits for it)
import statistics
st = {k: [] for k in ("tokenize", "embed (enqueue)", "search_knn (incl. wait)", "check + parse", "total")}
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inputs = col.encode_query([query], max_length=256); t1 = time.perf_counter()
    qv = dr.retriever.query(inputs).detach(); t2 = time.perf_counter()
    knn = ix.search_knn(query_vectors=qv, top_docs=10, index_batch_size=1024, verbose=False); t3 = time.perf_counter()
    dr._check_inputs(); out = dr.parse_indexer_output(knn); t4 = time.perf_counter()
    for k, v in zip(st, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)): st[k].append(v * 1e3)
print("  stages of the surface hop (median ms): " + " | ".join(f"{k} {statistics.median(v):.3f}" for k, v in st.items()))
# the same GPU work from tokens already on the device, stage by stage
st2 = {k: [] for k in ("forward (enqueue)", "search (incl. wait)", "total")}
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    qv = enc._hip.forward(ids, mask, 0); t1 = time.perf_counter()
    r = ix.index.search(qv, 10); t2 = time.perf_counter()
    for k, v in zip(st2, (t1 - t0, t2 - t1, t2 - t0)): st2[k].append(v * 1e3)
print("  stages of the C-ABI hop (median ms): " + " | ".join(f"{k} {statistics.median(v):.3f}" for k, v in st2.items()))
def med(fn, reps=200):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)
def manual():
    inputs = col.encode_query([query], max_length=256)
    qv = dr.retriever.query(inputs).detach()
    knn = ix.search_knn(query_vectors=qv, top_docs=10, index_batch_size=1024, verbose=False)
    dr._check_inputs(); return dr.parse_indexer_output(knn)
print(f"  variants (median of 200, sync before each): dr([q], 10) {med(lambda: dr([query], 10)):.3f} | dr.batch_retrieve {med(lambda: dr.batch_retrieve([query], topk=10)):.3f} | "
      f"dr._embed(on_device) + search_knn + parse {med(lambda: dr.parse_indexer_output(ix.search_knn(dr._embed([query], 'query', None, False, on_device=True), 10, verbose=False))):.3f} | manual pieces {med(manual):.3f} | "
      f"manual with max_length=None {med(lambda: dr.parse_indexer_output(ix.search_knn(dr.retriever.query(col.encode_query([query])).detach(), 10, verbose=False))):.3f}")
