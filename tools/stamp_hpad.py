#!/usr/bin/env python3
"""Stamp breakdown of FF1 / FF2 with the h row pitch padded (KIRAG_AMD_HPAD read at encoder creation; diagnostic build)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kirag_amd import _lib, bench_support as BS
lib = _lib.load(); raw = C.CDLL(_lib.LIB_PATH)
def read():
    buf = (C.c_ulonglong * 256)(); assert raw.kr_debug_read_stamps_enc(buf) == 0
    return np.array(buf[:], dtype=np.float64).reshape(8, 8, 4)
dev = torch.device("cuda:0")
names = {0: "QKV", 1: "out-proj", 2: "FF1+GELU", 3: "FF2"}
ref = {}
for pad in ("0", "64", "0", "64", "192"):
    os.environ["KIRAG_AMD_HPAD"] = pad
    enc = BS.make_hip_encoder(dev)
    for (B, S) in ((125, 32), (1000, 32), (1024, 128)):
        ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
        for _ in range(3): o = enc.forward(ids, mask, 0)
        torch.cuda.synchronize(); read(); t0 = time.perf_counter()
        reps = 10 if B * S < 20000 else 4
        for _ in range(reps): o = enc.forward(ids, mask, 0)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
        a = read()
        same = torch.equal(ref.setdefault((B, S), o), o)
        line = f"hpad={pad} {B}x{S}: {ms:.3f} ms same={same} |"
        for slot in (2, 3):
            x = a[slot]
            if x[:, 3].sum() == 0: continue
            for g, ws in (("g0", slice(0, 4)), ("g1", slice(4, 8))):
                t = x[ws, 3].sum()
                line += f" {names[slot]} {g}: loop {x[ws, 1].sum() / t / 1e3:.1f}k epi {x[ws, 2].sum() / t / 1e3:.1f}k |"
        print(line, flush=True)
    del enc
