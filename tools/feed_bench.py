#!/usr/bin/env python3
"""can the host keep the encoder fed?  Measures, on the GPU box,
  (a) the tokenizer alone (HF BertTokenizerFast, synthetic 30522-entry WordPiece vocab, ~110-token passages, batches of 512): one Rust thread,
      the tokenizer's own thread pool, N worker processes;
  (b) compute_corpus_embeddings.cal_doc_embeddings end to end at the e5-large shape (synthetic weights), prefetch_batches in {2, 8} and
      tokenizer_workers in {0, 8}, against the encoder's rate on pre-tokenised, device-resident batches.
Usage: python tools/feed_bench.py [passages]"""
import os, sys, tempfile, time
from types import SimpleNamespace
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000
td = tempfile.mkdtemp()
from kirag_amd.bench_support import synthetic_text_corpus
_, texts = synthetic_text_corpus(n, td)

def tok_rate(parallel, label):
    os.environ["TOKENIZERS_PARALLELISM"] = "true" if parallel else "false"
    from kirag_amd.bench_support import wordpiece_tokenizer
    tok = wordpiece_tokenizer(os.path.join(td, "vocab.txt"))
    tok(["passage: " + t for t in texts[:512]], max_length=128, padding=True, truncation=True, return_tensors="pt")
    t0 = time.perf_counter(); ntok = 0
    for s in range(0, n, 512):
        e = tok(["passage: " + t for t in texts[s:s + 512]], max_length=128, padding=True, truncation=True, return_tensors="pt")
        ntok += int(e["attention_mask"].sum())
    dt = time.perf_counter() - t0
    print(f"[tokenizer] {label}: {n / dt:.0f} passages/s ({ntok / n:.0f} tokens per passage)", flush=True)
    return tok

import subprocess
if os.environ.get("FEED_BENCH_CHILD") == "single":
    tok_rate(False, "one Rust thread (TOKENIZERS_PARALLELISM=false)"); sys.exit(0)
subprocess.run([sys.executable, __file__, str(n)], env={**os.environ, "FEED_BENCH_CHILD": "single"})     # the setting is read once per process
tok = tok_rate(True, f"tokenizer's own thread pool ({os.cpu_count()} cpus visible)")

import torch
from kirag_amd import bench_support as BS
from kirag_amd import compute_corpus_embeddings as CC
from kirag_amd.collators import E5Collator
dev = torch.device("cuda:0")
hip = BS.make_hip_encoder(dev)
class Model:
    encoder = SimpleNamespace(_hip=hip)
    def to(self, d): return self
    def eval(self): return self
    def doc(self, a): return hip.forward(a["input_ids"], a["attention_mask"], 0)
    def doc_packed(self, ids, lens, S, T=None): return hip.forward_packed(ids, lens, S, 0, T)
class Corpus:
    index_to_passage_id = {i: str(i) for i in range(n)}
    def __len__(self): return n
    def __getitem__(self, i): return {"index": i, "passage": texts[i]}
col = E5Collator(tokenizer=tok, query_maxlength=128, doc_maxlength=128)
# encoder alone on pre-tokenised resident batches
a = col.encode_doc(texts[:512]); ids = a["input_ids"].to(dev); mask = a["attention_mask"].to(dev)
hip.forward(ids, mask, 0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    hip.forward(ids, mask, 0)
torch.cuda.synchronize()
enc_rate = 20 * 512 / (time.perf_counter() - t0)
print(f"[encoder] pre-tokenised resident batches of 512: {enc_rate:.0f} passages/s", flush=True)
def run_loop(label, model, **kw):
    args = SimpleNamespace(local_rank=-1, save_dir=td, name="f", index_folder=label.replace(" ", "_"), per_gpu_batch_size=8, num_passage_per_index_file=10**6,
                           encode_batch_size=512, **kw)
    t0 = time.perf_counter()
    CC.cal_doc_embeddings(args, model, Corpus(), col, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lf = CC.cal_doc_embeddings.last_feed
    print(f"[end to end] {label}: {n / dt:.0f} passages/s = {n / dt / enc_rate * 100:.0f} % of the encoder-only rate   (loop {lf['loop_s']:.2f} s + tail {lf['tail_s']:.2f} s; "
          f"batches by producer {lf['batches_by_producer']})", flush=True)
    import shutil
    shutil.rmtree(os.path.join(td, "f", args.index_folder), ignore_errors=True)
run_loop("warm-up (process start, first allocations)", Model(), no_embedding_files=True)
run_loop("DEFAULT flags (tokenizer_workers=-1, prefetch_batches=2), shard files written", Model())
run_loop("default flags, no_embedding_files", Model(), no_embedding_files=True)
run_loop("tokenizer_workers=0, shard files written", Model(), tokenizer_workers=0)
for depth, workers in ((2, 0), (2, 2), (2, 4), (2, 8), (8, 8)):
    run_loop(f"prefetch_batches={depth} tokenizer_workers={workers} no_embedding_files", Model(), prefetch_batches=depth, tokenizer_workers=workers, no_embedding_files=True)
run_loop("padded_feed=True (the rounds 3-5 upload: int64 input_ids + attention_mask) tokenizer_workers=8 no_embedding_files", Model(), padded_feed=True,
         tokenizer_workers=8, no_embedding_files=True)

# the same loop through the MODULE surface the reference's callers use (E5Encoder.forward with input_ids / attention_mask / token_type_ids on the device):
# until round 3 its token_type_ids test synchronised the host on every forward (round 4: the types go to the kernels, nothing is tested on the host)
from transformers import BertConfig
from kirag_amd.retriever.encoders import E5Encoder
cfg = BertConfig(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, max_position_embeddings=512)
del hip
mod = E5Encoder(cfg, add_pooling_layer=False).to(dev).eval()
class ModelM:
    encoder = mod
    def to(self, d): return self
    def eval(self): return self
    def doc(self, a): return mod(**a)
    def doc_packed(self, ids, lens, S, T=None): return mod.forward_packed(ids, lens, S, T)
run_loop("module surface (E5Encoder.forward_packed), DEFAULT flags, shard files written", ModelM())
run_loop("module surface, padded_feed=True tokenizer_workers=8 (E5Encoder.forward on int64 input_ids / attention_mask)", ModelM(), padded_feed=True, tokenizer_workers=8,
         no_embedding_files=True)
