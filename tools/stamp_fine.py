#!/usr/bin/env python3
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kirag_amd import _lib, bench_support as BS
lib = _lib.load(); raw = C.CDLL(_lib.LIB_PATH)
def read():
    buf = (C.c_ulonglong * 256)(); assert raw.kr_debug_read_stamps_enc(buf) == 0
    fb = (C.c_ulonglong * 128)(); assert raw.kr_debug_read_fine_enc(fb) == 0
    return np.array(buf[:], dtype=np.float64).reshape(8, 8, 4), np.array(fb[:], dtype=np.float64).reshape(8, 2, 8)
dev = torch.device("cuda:0"); enc = BS.make_hip_encoder(dev)
names = {0: "QKV", 1: "out-proj", 2: "FF1+GELU", 3: "FF2"}
for (B, S) in ((125, 32), (1000, 32)):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    for _ in range(3): enc.forward(ids, mask, 0)
    torch.cuda.synchronize(); read()
    for _ in range(5): enc.forward(ids, mask, 0)
    torch.cuda.synchronize(); a, f = read()
    for slot, nm in names.items():
        x = a[slot]
        if x[:, 3].sum() == 0: continue
        for g, ws in ((0, slice(0, 4)), (1, slice(4, 8))):
            t = x[ws, 3].sum()
            print(f"{B}x{S} {nm:10s} g{g}: epi {x[ws, 2].sum() / t / 1e3:6.1f}k | steps (write0, +read/write1, store0 ..): " + " ".join(f"{v / t / 1e3:6.1f}k" for v in f[slot, g, :6]), flush=True)
