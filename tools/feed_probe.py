#!/usr/bin/env python3
"""Where does the corpus-encode feed lose time on this host?  For a few feed configurations: end-to-end rate of cal_doc_embeddings from text, the cgroup's CPU
throttling counters around each run (a 16-CPU quota on a 256-CPU machine: 256-thread pools get throttled), who produced the batches, and how long the consumer
waited for frames.  Usage: python tools/feed_probe.py [passages] ; environment RAYON_NUM_THREADS is honoured by the tokenizer's pool."""
import os, sys, tempfile, time
from types import SimpleNamespace
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

def cpu_stat():
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except OSError:
            continue
    return out

def quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return "unlimited" if q == "max" else f"{int(q) / int(p):.1f} cpus"
    except OSError:
        return "?"

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
print(f"cpus visible {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, cgroup quota {quota()}, RAYON_NUM_THREADS={os.environ.get('RAYON_NUM_THREADS')}", flush=True)
import torch
from kirag_amd import bench_support as BS
from kirag_amd import compute_corpus_embeddings as CC
from kirag_amd import feed as F
from kirag_amd.collators import E5Collator
td = tempfile.mkdtemp()
vocab, texts = BS.synthetic_text_corpus(n, td)
tok = BS.wordpiece_tokenizer(vocab)
col = E5Collator(tokenizer=tok, query_maxlength=128, doc_maxlength=128)
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
a = col.encode_doc(texts[:512]); ids = a["input_ids"].to(dev); mask = a["attention_mask"].to(dev)
hip.forward(ids, mask, 0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    hip.forward(ids, mask, 0)
torch.cuda.synchronize()
enc_rate = 20 * 512 / (time.perf_counter() - t0)
print(f"[encoder] {enc_rate:.0f} passages/s", flush=True)

# consumer wait time: wrap TokenFeed.__iter__
orig_iter = F.TokenFeed.__iter__
def timed_iter(self):
    self.wait_s = 0.0
    it = orig_iter(self)
    while True:
        t0 = time.perf_counter()
        try:
            fr = next(it)
        except StopIteration:
            return
        self.wait_s += time.perf_counter() - t0
        yield fr
F.TokenFeed.__iter__ = timed_iter
feeds = []
orig_init = F.TokenFeed.__init__
def init(self, *a, **k):
    if "FORCE_LOCAL" in os.environ:
        k["local"] = os.environ["FORCE_LOCAL"] == "1"
    orig_init(self, *a, **k); feeds.append(self)
F.TokenFeed.__init__ = init

def run(label, **kw):
    args = SimpleNamespace(local_rank=-1, save_dir=td, name="f", index_folder="x", per_gpu_batch_size=8, num_passage_per_index_file=10**6, encode_batch_size=512, **kw)
    s0 = cpu_stat(); t0 = time.perf_counter()
    CC.cal_doc_embeddings(args, Model(), Corpus(), col, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0; s1 = cpu_stat()
    lf = CC.cal_doc_embeddings.last_feed
    thr = {k: s1.get(k, 0) - s0.get(k, 0) for k in ("nr_throttled", "throttled_usec", "usage_usec")}
    print(f"[{label}] {n / dt:.0f} passages/s = {n / dt / enc_rate * 100:.0f} %; consumer waited {feeds[-1].wait_s:.2f} s of {dt:.2f} s for frames; producers {lf['batches_by_producer']}; "
          f"cpu used {thr['usage_usec'] / 1e6:.1f} cpu-s, throttled {thr['nr_throttled']} periods / {thr['throttled_usec'] / 1e6:.2f} s", flush=True)
    import shutil; shutil.rmtree(os.path.join(td, "f"), ignore_errors=True)

run("warm", tokenizer_workers=0, no_embedding_files=True)
run("workers=0 no files", tokenizer_workers=0, no_embedding_files=True)
run("workers=0 files (streamed)", tokenizer_workers=0)
run("workers=0 files (buffered + pickle.dump)", tokenizer_workers=0, buffered_shard_files=True)
run("DEFAULT flags", )
run("workers=4 no files", tokenizer_workers=4, no_embedding_files=True)
os.environ["FORCE_LOCAL"] = "0"
run("workers=4 only (no local thread) no files", tokenizer_workers=4, no_embedding_files=True)
del os.environ["FORCE_LOCAL"]
