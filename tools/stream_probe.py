import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
import numpy as np, torch
from kirag_amd import bench_support as BS, compute_corpus_embeddings as CC
from kirag_amd.collators import E5Collator
from kirag_amd.retriever.index import Indexer
n = 32768
td = tempfile.mkdtemp()
vocab, texts = BS.synthetic_text_corpus(n, td)
col = E5Collator(tokenizer=BS.wordpiece_tokenizer(vocab), query_maxlength=128, doc_maxlength=128)
dev = torch.device("cuda:0")
hip = BS.make_hip_encoder(dev)
class Model:
    encoder = SimpleNamespace(_hip=hip)
    def to(self, d): return self
    def eval(self): return self
    def doc(self, a): return hip.forward(a["input_ids"], a["attention_mask"], 0)
    def doc_packed(self, ids, lens, S, T=None): return hip.forward_packed(ids, lens, S, 0, T)
class Corpus:
    index_to_passage_id = {i: str(10_000_000 + i) for i in range(n)}
    def __len__(self): return n
    def __getitem__(self, i): return {"index": i, "passage": texts[i]}
def run(label, indexer):
    args = SimpleNamespace(local_rank=-1, save_dir=td, name="f", index_folder=label, per_gpu_batch_size=8, num_passage_per_index_file=10**6, encode_batch_size=512, no_embedding_files=True)
    t0 = time.perf_counter(); CC.cal_doc_embeddings(args, Model(), Corpus(), col, device=dev, indexer=indexer); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{label}] {n / dt:.0f} passages/s" + (f", index rows {indexer.index.ntotal}, byte_rows {indexer.index.stats()['byte_rows']}" if indexer else ""), flush=True)
run("warm", None)
run("no indexer", None)
small = Indexer(1024); run("fresh resident index (below 2^19 rows: no prepare)", small)
big = Indexer(1024)
g = torch.Generator(device=dev); g.manual_seed(1)
for s0 in range(0, 600_000, 100_000):
    big.index.add(torch.nn.functional.normalize(torch.randn(100_000, 1024, device=dev, generator=g), dim=1))
big.index_id_to_db_id = np.arange(600_000, dtype=np.int64)
run("appended to a 600 k-row index (kr_index_prepare after every batch)", big)
