#!/usr/bin/env python3
"""Stamp breakdown of the projection kernels for small batches under KIRAG_AMD_STORE_NT=0/1 (diagnostic build)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kirag_amd import _lib, bench_support as BS
lib = _lib.load(); raw = C.CDLL(_lib.LIB_PATH)
def read():
    buf = (C.c_ulonglong * 256)(); assert raw.kr_debug_read_stamps_enc(buf) == 0
    return np.array(buf[:], dtype=np.float64).reshape(8, 8, 4)
dev = torch.device("cuda:0"); enc = BS.make_hip_encoder(dev)
names = {0: "QKV", 1: "out-proj", 2: "FF1+GELU", 3: "FF2"}
for (B, S) in ((125, 32), (250, 32), (500, 32)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    for nt in ("1", "0"):
        os.environ["KIRAG_AMD_STORE_NT"] = nt
        for _ in range(3): enc.forward(ids, mask, 0)
        torch.cuda.synchronize(); read(); t0 = time.perf_counter()
        for _ in range(5): enc.forward(ids, mask, 0)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        a = read()
        line = f"{B}x{S} nt={nt}: {ms:.3f} ms |"
        for slot, nm in names.items():
            x = a[slot]
            if x[:, 3].sum() == 0: continue
            for g, ws in (("g0", slice(0, 4)), ("g1", slice(4, 8))):
                t = x[ws, 3].sum()
                line += f" {nm} {g}: loop {x[ws, 1].sum() / t / 1e3:.1f}k epi {x[ws, 2].sum() / t / 1e3:.1f}k |"
        print(line, flush=True)
