#!/usr/bin/env python3
"""In-kernel cycle breakdown of the ping-pong GEMM loop (diagnostic build: tools/stamp_build.sh; run with
KIRAG_AMD_LIB=tools/bin/libkirag_amd_stamp.so python tools/stamp_run.py).  Prints, per kernel kind and wave group, cycles per output tile spent
in tile set-up, in the K loop and in the epilogue."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kirag_amd import _lib, bench_support as BS
from kirag_amd.bench_support import CorpusDist
from kirag_amd.retriever.index import FlatIPIndex
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)

def read(fn):
    buf = (C.c_ulonglong * 256)()
    assert getattr(raw, fn)(buf) == 0
    return np.array(buf[:], dtype=np.float64).reshape(8, 8, 4)

def show(tag, a, names):
    for slot, nm in names.items():
        x = a[slot]
        if x[:, 3].sum() == 0:
            continue
        for g, ws in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
            t = x[ws, 3].sum()
            print(f"[{tag}] {nm:18s} {g}: tiles/wave {t / 4:.0f}  setup {x[ws, 0].sum() / t:8.0f}  K loop {x[ws, 1].sum() / t:8.0f}  epilogue {x[ws, 2].sum() / t:8.0f}  total {x[ws, :3].sum() / t:8.0f} cycles per tile")

dev = torch.device("cuda:0")
n, nq, d, k = 2_000_000, 1000, 1024, 100
cd = CorpusDist("gaussian", d, dev); g = torch.Generator(device=dev); g.manual_seed(3)
ix = FlatIPIndex(d, device=0); ix.reserve(n)
head = None
for s0 in range(0, n, 250_000):
    x = cd.rows(250_000, g); ix.add(x); head = x[:nq].clone() if head is None else head
q = cd.queries_near(head, torch.Generator(device=dev).manual_seed(2))
for _ in range(3):
    ix.search(q, k)
read("kr_debug_read_stamps")
for _ in range(5):
    ix.search(q, k)
print("coarse ms", ix.stats()["last_coarse_ms"])
show("search", read("kr_debug_read_stamps"), {0: "k_coarse"})
del ix
enc = BS.make_hip_encoder(dev)
for (B, S) in ((1024, 128), (1000, 32), (125, 32)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    for _ in range(2):
        enc.forward(ids, mask, 0)
    torch.cuda.synchronize(); read("kr_debug_read_stamps_enc")
    import time
    t0 = time.perf_counter()
    for _ in range(3):
        enc.forward(ids, mask, 0)
    torch.cuda.synchronize()
    print(f"encoder {B} x {S}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per forward")
    show(f"enc {B}x{S}", read("kr_debug_read_stamps_enc"), {0: "QKV (K=1024)", 1: "out-proj (K=1024)", 2: "FF1+GELU (K=1024)", 3: "FF2 (K=4096)"})
