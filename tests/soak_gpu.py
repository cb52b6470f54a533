#!/usr/bin/env python3
"""Randomised soak of the HIP paths on one GPU (python tests/soak_gpu.py [seconds] [seed]; not collected by pytest): every case is checked on the device itself —
  * index: certified MFMA scan (mode 0, every coarse kernel variant: <= 32 / <= 128 / 1024-query blocks, 1..4 rounds, incremental adds, k up to 1024,
    clustered / near-duplicate data that sends queries to pass 2) must equal the exact scan (mode 1) AND the fp64-MFMA pass alone (mode 2) in rows and
    score bits; every fourth case with every canonical score forced through the integer super-accumulator; every sixth case with non-unit row norms;
    a sample of scores against torch's fp32 matmul (independent arithmetic);
  * index, every fourth case (round 5): the corpus cut into 2-3 uneven row shards searched in SPLIT form (coarse scan -> gathered coarse scores -> global bound
    -> re-rank above it), the merged lists must equal the unsharded search bit for bit;
  * index (round 5): the byte pre-scan of small query blocks (kr_set_option "byte_prescan") is enabled for EVERY index size here (debug_byte_min_rows = 0), so
    blocks of <= 32 queries at d in {512, 768, 1024} take it whenever their last round is not the direct one; every seventh case plants a NaN row (the int8
    copy then marks every row: slow, still exact); every fifth case adds rows AFTER the first search (the int8 copy is extended) and searches again;
  * index (round 6): every third case calls kr_index_prepare(nq, k) before its first search (the int8 copy and the workspaces exist up front; results must not change);
  * encoder (round 6): when every mask of a case is right-padded... every case also runs the RAGGED forward (kr_encoder_forward_packed) on the right-padded twin
    of its batch, which must equal the padded forward of that twin bit for bit;
  * encoder (tiny config): the projection main loops / skinny tile shapes (KIRAG_AMD_PROJ_TILE = 256 / 130 / 128 / 64 / 32) must agree bit for bit, and a
    sequence's embedding must not depend on the rest of the batch.
Exits non-zero on the first mismatch; prints a progress line every 30 s (the GPU box kills a silent run) and ONE summary line at the end: profiles/rNN keeps the
summary, not the log."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
from kirag_amd import _lib
from kirag_amd.retriever.index import FlatIPIndex
from kirag_amd.retriever.encoders import HipBertForward

t_end = time.time() + budget
cases = 0
byte_scans = 0
last_print = time.time()
tot = {"certified": 0, "fine": 0, "exact": 0}
_lib.check(_lib.load().kr_set_option(b"debug_byte_min_rows", 0))


def unit(n, d):
    x = torch.randn(n, d, device="cuda")
    return torch.nn.functional.normalize(x, dim=1)


# ---- index -------------------------------------------------------------------------------------------------------------------------------
while time.time() < t_end - budget * 0.35:
    d = int(rng.choice([64, 128, 384, 512, 768, 1024]))
    n = int(rng.choice([1, 7, 100, 1000, 5000, 33333, 120000, 300000])) + int(rng.integers(0, 50))
    nq = int(rng.choice([1, 2, 5, 31, 32, 33, 100, 128, 129, 300, 1000]))
    k = int(min(n, rng.choice([1, 3, 10, 20, 100, 256, 600, 1024])))
    ix = FlatIPIndex(d, device=0)
    x = unit(n, d)
    if cases % 5 == 1 and n > 50:                       # a tight cluster (half of the rows within ~1e-4 of one direction): pass 1 cannot certify there
        x[: n // 2] = torch.nn.functional.normalize(x[0] + float(rng.choice([1e-3, 1e-4, 2e-5])) * torch.randn(n // 2, d, device="cuda"), dim=1)
    if cases % 6 == 2:
        x = x * float(rng.choice([0.01, 3.0, 250.0]))
    if n > 10 and rng.random() < 0.5:
        cut = int(rng.integers(1, n))
        ix.add(x[:cut]); ix.add(x[cut:])
    else:
        ix.add(x)
    if n > 20:   # duplicates: ties by row
        x2 = x.clone(); x2[n - 3] = x2[1]
        if cases % 7 == 5 and k < n:               # (k == n would have to return the NaN row itself)
            x2[n // 2] = float("nan")               # a NaN row is never returned; the byte pre-scan gives up on such an index after marking every row once
        ix = FlatIPIndex(d, device=0); ix.add(x2); x = x2
    pick = torch.from_numpy(rng.integers(0, n, nq)).cuda()
    pick[pick == n // 2] = 0                            # (never the planted NaN row)
    q = torch.nn.functional.normalize(x[pick] + 0.3 * torch.randn(nq, d, device="cuda") / d ** 0.5, dim=1)
    if cases % 3 == 2:
        ix.prepare(min(nq, 1024), k)                    # round 6: everything the first search would set up, done ahead of it
    force = cases % 4 == 3
    if force:
        _lib.check(_lib.load().kr_set_option(b"force_exact_scores", 1))
    s0, i0 = ix.search(q, k)
    if force:
        _lib.check(_lib.load().kr_set_option(b"force_exact_scores", 0))
    nq1 = min(nq, 8)                                    # the per-query exact scan is slow by design: a few queries
    s1, i1 = ix.search(q[:nq1], k, mode=1)
    why = []
    ok = np.array_equal(i0[:nq1], i1) and np.array_equal(s0[:nq1].view(np.uint32), s1.view(np.uint32))
    if not ok:
        why.append(("mode0 != mode1", int((i0[:nq1] != i1).sum()), int((s0[:nq1].view(np.uint32) != s1.view(np.uint32)).sum())))
    if d <= 2048:
        s2, i2 = ix.search(q, k, mode=2)
        ok2 = np.array_equal(i0, i2) and np.array_equal(s0.view(np.uint32), s2.view(np.uint32))
        if not ok2:
            bad = np.nonzero((i0 != i2).any(1) | (s0.view(np.uint32) != s2.view(np.uint32)).any(1))[0]
            why.append(("mode0 != mode2", bad[:5].tolist(), i0[bad[0]][:4].tolist(), i2[bad[0]][:4].tolist(), s0[bad[0]][:4].tolist(), s2[bad[0]][:4].tolist(), ix.stats()))
        ok = ok and ok2
    if cases % 3 == 0:                                  # several asynchronous searches outstanding on one stream (ABI 5), finished together
        m = int(rng.integers(2, 5))
        qs = [torch.roll(q, r, dims=0).contiguous() for r in range(m)]
        outs_a = [(torch.empty((nq, k), dtype=torch.float32, device="cuda"), torch.empty((nq, k), dtype=torch.int64, device="cuda")) for _ in range(m)]
        for qq, (sa, ia) in zip(qs, outs_a):
            ix.search_async(qq, k, sa, ia)
        fl = ix.finish()
        oka = len(fl) == m
        for r, (sa, ia) in enumerate(outs_a):
            oka = oka and np.array_equal(ia.cpu().numpy(), np.roll(i0, r, axis=0)) and np.array_equal(sa.cpu().numpy().view(np.uint32), np.roll(s0, r, axis=0).view(np.uint32))
        if not oka:
            why.append(("outstanding async searches != blocking search", m, fl))
        ok = ok and oka
    if cases % 4 == 1 and n >= 2 * k + 2 and nq <= 1024:   # round 5: the row-sharded SPLIT search (exchange of coarse scores before the re-rank), 2-3 uneven shards of this corpus
        W = int(rng.integers(2, 4))
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(k, n - k + 1, W - 1)]))
        if all(b - a >= k for a, b in zip(cuts[:-1], cuts[1:])):
            shards = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                sh = FlatIPIndex(d, device=0); sh.add(x[a:b]); shards.append((sh, a))
            tks = [torch.empty((nq, k + 1), dtype=torch.float32, device="cuda") for _ in shards]
            for (sh, a), tk in zip(shards, tks):
                sh.search_coarse_async(q, k, tk)
            gathered = torch.cat(tks, dim=0).contiguous()
            sc_all, id_all = [], []
            for (sh, a) in shards:
                th = torch.empty((nq,), dtype=torch.float32, device="cuda")
                sc = torch.empty((nq, k), dtype=torch.float32, device="cuda"); rw = torch.empty((nq, k), dtype=torch.int64, device="cuda")
                sh.search_global_theta(gathered, len(shards), th); sh.search_rerank_async(th, sc, rw); sh.finish()
                r_ = rw.cpu().numpy(); sc_all.append(sc.cpu().numpy()); id_all.append(np.where(r_ >= 0, r_ + a, -1))
            ms = np.empty((nq, k), np.float32); mi = np.empty((nq, k), np.int64)
            sc_st, id_st = np.ascontiguousarray(np.stack(sc_all)), np.ascontiguousarray(np.stack(id_all))      # named: they must outlive the call
            _lib.check(_lib.load().kr_topk_merge(sc_st.ctypes.data, id_st.ctypes.data, len(shards), nq, k, ms.ctypes.data, mi.ctypes.data))
            oks = np.array_equal(mi, i0) and np.array_equal(ms.view(np.uint32), s0.view(np.uint32))
            if not oks:
                bad = np.nonzero((mi != i0).any(1))[0]
                why.append(("split sharded search != unsharded", cuts, bad[:5].tolist()))
            ok = ok and oks
            del shards
    if cases % 5 == 4 and nq <= 8 and n > 1000:        # rows added AFTER the first search: the int8 copy is extended (or rebuilt), the new best rows must be found
        extra = torch.nn.functional.normalize(q[:1] + 0.05 * torch.randn(300, d, device="cuda") / d ** 0.5, dim=1) * float(x[1].norm())
        ix.add(extra)
        se, ie = ix.search(q, k); se1, ie1 = ix.search(q, k, mode=1)
        oke = np.array_equal(ie, ie1) and np.array_equal(se.view(np.uint32), se1.view(np.uint32)) and (ie[0, : min(k, 5)] >= n).sum() >= min(k, 5) - 1   # (the picked row itself may still lead)
        if not oke:
            why.append(("after a later add: mode0 != mode1", ie[0, :5].tolist(), ie1[0, :5].tolist()))
        ok = ok and oke
    nan_rows = torch.isnan(x).any(1)
    ref = (q[:4] @ torch.where(nan_rows[:, None], torch.zeros_like(x), x).T).cpu().numpy()   # independent arithmetic: fp32 matmul scores of the returned rows
    got = np.take_along_axis(ref, i0[:4], axis=1)
    scale = float(np.abs(ref).max()) + 1e-30
    ok3 = np.abs(got - s0[:4]).max() <= 4e-6 * scale and (np.diff(s0, axis=1) <= 0).all()
    if not ok3:
        why.append(("fp32 matmul / order", float(np.abs(got - s0[:4]).max()), scale, i0[:4, :3].tolist(), pick[:4].tolist(), s0[:4, :3].tolist(), got[:, :3].tolist(),
                    (q[:4] * x[torch.from_numpy(i0[:4, 0]).cuda()]).sum(1).tolist(), float((ix.reconstruct_n(int(i0[0, 0]), 1)[0] - x[int(i0[0, 0])].cpu().numpy()).max())))
    ok = ok and ok3
    if not ok:
        print("INDEX MISMATCH", dict(n=n, d=d, nq=nq, k=k, seed=seed, case=cases, force=force), why, flush=True)
        sys.exit(1)
    cases += 1
    byte_scans += ix.stats()["byte_scans"]
    tot["certified"] += ix.stats()["certified"]; tot["fine"] += ix.stats()["fine"]; tot["exact"] += ix.stats()["exact"]
    if time.time() - last_print > 30:
        last_print = time.time()
        print(f"[stress] {cases} index cases ok, {byte_scans} blocks through the byte pre-scan (last case n={n} d={d} nq={nq} k={k})", flush=True)
    del ix, x
index_cases = cases

# ---- encoder -----------------------------------------------------------------------------------------------------------------------------
from oracle import encoder_np as E   # synthetic-weights generator (this file lives under tests/: the oracle is test infrastructure)
cfg = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                      max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
w = E.synth_weights(128, 2, 512, 1000, 512, seed=11)
enc = HipBertForward(cfg, 0)
enc.load_state(w)
while time.time() < t_end:
    B = int(rng.choice([1, 2, 3, 8, 17, 40, 130])); S = int(rng.choice([1, 5, 32, 33, 64, 100, 129, 200, 300]))
    ids = rng.integers(5, 1000, (B, S)); mask = np.zeros((B, S), np.int64)
    lens = rng.integers(1, S + 1, B); lens[0] = S
    for b in range(B):
        if b % 3 == 2: mask[b, S - lens[b]:] = 1          # left padded
        else: mask[b, :lens[b]] = 1
    pool = int(rng.integers(0, 2))
    outs = []
    for tile in ("256", "130", "128", "64", "32"):
        os.environ["KIRAG_AMD_PROJ_TILE"] = tile
        outs.append(enc.forward_np(ids, mask, pool))
    os.environ.pop("KIRAG_AMD_PROJ_TILE")
    auto = enc.forward_np(ids, mask, pool)
    alone = enc.forward_np(ids[:1], mask[:1], pool)           # sequence 0 on its own: batch independence
    ok = all(np.array_equal(o.view(np.uint32), outs[0].view(np.uint32)) for o in outs[1:] + [auto]) and np.array_equal(alone[0].view(np.uint32), auto[0].view(np.uint32))
    if not ok or not np.isfinite(auto).all():
        print("ENCODER MISMATCH", dict(B=B, S=S, pool=pool, seed=seed, case=cases), flush=True)
        sys.exit(1)
    # the ragged forward on the right-padded twin of this batch (same lengths, every sequence moved to the left edge)
    rmask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    rids = np.zeros_like(ids)
    for b in range(B):
        rids[b, :lens[b]] = ids[b][mask[b] != 0]
    padded = enc.forward_np(rids, rmask, pool)
    rag = np.ascontiguousarray(rids[rmask != 0].astype(np.int32))
    packed = enc.forward_packed(torch.from_numpy(rag), torch.from_numpy(lens.astype(np.int32)), S, pool).cpu().numpy()
    enc.check()
    if not np.array_equal(packed.view(np.uint32), padded.view(np.uint32)):
        print("ENCODER MISMATCH (packed forward)", dict(B=B, S=S, pool=pool, seed=seed, case=cases), flush=True)
        sys.exit(1)
    cases += 1
    if time.time() - last_print > 30:
        last_print = time.time()
        print(f"[stress] {cases} cases ok (last encoder case B={B} S={S} pool={pool})", flush=True)
print(f"[stress] done: seed {seed}, {budget:.0f} s, {cases} cases ({index_cases} index: {tot['certified']} queries certified / {tot['fine']} pass 2 / {tot['exact']} pass 3, "
      f"{byte_scans} blocks through the byte pre-scan; {cases - index_cases} encoder incl. the packed forward), no mismatch", flush=True)
