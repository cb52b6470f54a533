#!/usr/bin/env python3
"""A/B of the LayerNorm-in-prologue experiment (KIRAG_AMD_FUSE_LN=1: forwards of at most 32 packed tokens run 5 launches per layer instead of 7):
(1) parity: embeddings and last_hidden_state must equal the unfused forward BIT FOR BIT (tiny config and the full e5-large shape, both pools, ragged batches);
(2) latency of the forwards the fusion applies to, the two modes alternating on one box.  Usage: python tools/fuse_ln_ab.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from types import SimpleNamespace
import numpy as np
import torch
from kirag_amd import bench_support as BS
from kirag_amd.retriever.encoders import HipBertForward
from oracle import encoder_np as E          # synthetic weights only (this is a tool, not the product)

dev = torch.device("cuda:0")
def mode(on):
    if on: os.environ["KIRAG_AMD_FUSE_LN"] = "1"
    else: os.environ.pop("KIRAG_AMD_FUSE_LN", None)

def parity(enc, vocab, label):
    rng = np.random.default_rng(3)
    bad = 0
    for B, S in ((1, 32), (1, 7), (3, 10), (2, 16), (4, 8), (1, 1), (5, 6), (32, 1)):
        lens = rng.integers(1, S + 1, B); lens[0] = S
        ids = rng.integers(5, vocab, (B, S)); mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
        if B > 2:
            mask[1] = 0; mask[1, S - lens[1]:] = 1                      # one left-padded sequence
        for pool in (0, 1):
            mode(False); a = enc.forward_np(ids, mask, pool); ha = enc.last_hidden(B, S).numpy() if pool == 0 else None
            mode(True); b = enc.forward_np(ids, mask, pool); hb = enc.last_hidden(B, S).numpy() if pool == 0 else None
            ok = np.array_equal(a.view(np.uint32), b.view(np.uint32)) and (ha is None or np.array_equal(ha.view(np.uint32), hb.view(np.uint32)))
            bad += not ok
            if not ok:
                print(f"  MISMATCH {label} B={B} S={S} pool={pool}: max |diff| {np.nanmax(np.abs(a - b)):.3e}", flush=True)
    mode(False)
    print(f"[parity] {label}: {'bit-identical on every case' if bad == 0 else str(bad) + ' cases differ'}", flush=True)
    return bad == 0

cfg = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                      max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
tiny = HipBertForward(cfg, 0); tiny.load_state(E.synth_weights(128, 2, 512, 1000, 512, seed=11))
ok = parity(tiny, 1000, "tiny (H = 128, 2 layers)")
big = BS.make_hip_encoder(dev)
ok = parity(big, 30000, "e5-large shape (H = 1024, 24 layers)") and ok

def timed(B, S, reps=200):
    ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
    res = {}
    for rnd in range(3):
        for on in (False, True):
            mode(on)
            for _ in range(5): big.forward(ids, mask, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): big.forward(ids, mask, 0)
            torch.cuda.synchronize(); res.setdefault(on, []).append((time.perf_counter() - t0) / reps * 1e3)
    mode(False)
    return res
for B, S in ((1, 32), (1, 16), (2, 16), (4, 8)):
    r = timed(B, S)
    print(f"[latency] {B} x {S} tokens: 7 launches per layer {np.median(r[False]):.3f} ms ({' '.join('%.3f' % v for v in r[False])}), "
          f"5 launches per layer {np.median(r[True]):.3f} ms ({' '.join('%.3f' % v for v in r[True])})", flush=True)
r = timed(1, 64)      # 33+ tokens: the fusion does not apply, both modes run the same launches
print(f"[latency] 1 x 64 tokens (fusion not applicable): {np.median(r[False]):.3f} vs {np.median(r[True]):.3f} ms", flush=True)
sys.exit(0 if ok else 1)
