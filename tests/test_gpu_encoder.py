"""GPU parity tests of the encoder path (through the C ABI) against vectors produced by the reference's own
E5Encoder / BGEEncoder (tests/golden, tests/golden/make_golden.py) and against the numpy oracle.

Tolerances.  The reference computes in fp32; the HIP path feeds bf16 operands to the MFMAs (fp32 accumulate,
fp32 residual stream).  north_star: cosine scores within 1e-3.  Bars used here, on unit-norm outputs:
  * element-wise |out - ref| <= 4e-3 (tiny configs) / 3e-3 (full size),  1 - cos(out, ref) <= 5e-5
  * inner-product scores between encoded queries and passages within 1e-3 of the fp32 reference scores."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import encoder_np as E

pytestmark = pytest.mark.gpu


def _cfg(arr):
    H, L, heads, FF, vocab, max_pos = [int(v) for v in arr]
    return SimpleNamespace(hidden_size=H, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=FF, vocab_size=vocab,
                           max_position_embeddings=max_pos, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")


def _hip(cfg, weights):
    from kirag_amd.retriever.encoders import HipBertForward
    h = HipBertForward(cfg, 0)
    h.load_state(weights)
    return h


def _check(out, ref, atol, tag=""):
    assert out.shape == ref.shape
    err = np.abs(out - ref).max()
    cos = (out * ref).sum(1) / (np.linalg.norm(out, axis=1) * np.linalg.norm(ref, axis=1))
    assert err <= atol, f"{tag}: max abs err {err:.2e} > {atol}"
    assert (1 - cos).max() <= 5e-5, f"{tag}: 1-cos {float((1 - cos).max()):.2e}"
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
    return err


@pytest.mark.parametrize("name", ["t128", "t256"])
def test_g1_tiny_configs_match_reference(golden, name):
    g = golden("g1_encoder_tiny.npz")
    cfg = _cfg(g[f"cfg.{name}"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    h = _hip(cfg, w)
    worst = 0.0
    for tag, pool in (("e5", 0), ("bge", 1)):
        for ci in range(7):
            key = f"{name}.{tag}.c{ci}"
            out = h.forward_np(g[key + ".ids"], g[key + ".mask"], pool)
            worst = max(worst, _check(out, g[key + ".out"], 4e-3, key))
    print(f"[{name}] worst abs err vs reference fp32: {worst:.2e}")


@pytest.mark.parametrize("residual_lo,tol", [("0", 4e-2), ("1", 3e-2)])
def test_g1_last_hidden_state(golden, residual_lo, tol, monkeypatch):
    """last_hidden_state of the tiny config; LayerNorm outputs are O(1) (|x| up to ~4: one bf16 ulp is 0.016-0.03).  Default build: the residual
    stream between layers is bf16 and only the final LayerNorm keeps 16 mantissa bits; KIRAG_AMD_RESIDUAL_LO=1 (read at encoder creation) keeps
    them in every layer."""
    monkeypatch.setenv("KIRAG_AMD_RESIDUAL_LO", residual_lo)
    g = golden("g1_encoder_tiny.npz")
    cfg = _cfg(g["cfg.t256"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    h = _hip(cfg, w)
    for ci in (1, 3):
        key = f"t256.e5.c{ci}"
        ids, mask, hid = g[key + ".ids"], g[key + ".mask"], g[key + ".hidden"]
        h.forward_np(ids, mask, 0)
        lh = h.last_hidden(*ids.shape).numpy()
        keep = mask.astype(bool)
        assert np.abs(lh[keep] - hid[-1][keep]).max() <= tol
        assert (lh[~keep] == 0).all()


@pytest.mark.parametrize("residual_lo", ["0", "1"])
def test_g2_full_size_e5_and_bge(golden, residual_lo, monkeypatch):
    """Full-size (24-layer) goldens in both residual-stream modes (see encoder.hip header): the error is set by the bf16 GEMM operands."""
    monkeypatch.setenv("KIRAG_AMD_RESIDUAL_LO", residual_lo)
    g = golden("g2_encoder_large.npz")
    cfg = _cfg(g["cfg"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    h = _hip(cfg, w)
    outs = {}
    for key, pool in (("e5.c0", 0), ("e5.c1", 0), ("e5.c2", 0), ("bge.c0", 1)):
        out = h.forward_np(g[key + ".ids"], g[key + ".mask"], pool)
        err = _check(out, g[key + ".out"], 3e-3, key)
        outs[key] = out
        print(f"[{key}] max abs err {err:.2e}")
    # scores: queries (c2: 32-token) x passages (c1), vs the fp32 reference scores   (north_star: within 1e-3)
    s_hip = outs["e5.c2"] @ outs["e5.c1"].T
    s_ref = g["e5.c2.out"] @ g["e5.c1.out"].T
    print(f"[scores] max |q.d - ref| = {float(np.abs(s_hip - s_ref).max()):.2e}")
    assert np.abs(s_hip - s_ref).max() <= 1e-3, float(np.abs(s_hip - s_ref).max())


def _g10_spec():
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("g10_spec", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_spec.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


G10_RESULTS = {}
_G10_WEIGHTS = {}


def _g10_weights(spec, wname):
    if wname not in _G10_WEIGHTS:
        _G10_WEIGHTS.clear()                       # one 1.3-GB weight set at a time
        _G10_WEIGHTS[wname] = spec.weights(wname)
    return _G10_WEIGHTS[wname]


def _g10_run(golden, wname, dtype, lo):
    """-> (worst |q.d - ref| over all e5 x e5 case pairs, worst abs err, worst 1 - cos) of operand type `dtype` / residual low half `lo` on weight set `wname`"""
    from kirag_amd.retriever.encoders import HipBertForward
    spec = _g10_spec()
    g = golden("g10_encoder_large_ext.npz")
    cfg = _cfg(g["cfg"])
    h = HipBertForward(cfg, 0, operand_dtype=dtype, residual_lo=lo)
    assert h.operand_dtype == dtype and h.residual_lo == lo
    h.load_state(_g10_weights(spec, wname))
    outs, refs = {}, {}
    worst_err, worst_cos = 0.0, 0.0
    for tag, pool in (("e5", 0), ("bge", 1)):
        for ci, (B, S, layout, seed) in enumerate(spec.CASES[wname][tag]):
            ids, mask = spec.tokens(B, S, layout, seed)
            key = f"{wname}.{tag}.c{ci}"
            out = h.forward_np(ids, mask, pool)
            ref = g[key + ".out"]
            assert out.shape == ref.shape and np.isfinite(out).all(), key
            cos = (out * ref).sum(1) / (np.linalg.norm(out, axis=1) * np.linalg.norm(ref, axis=1))
            err = float(np.abs(out - ref).max())
            worst_err = max(worst_err, err); worst_cos = max(worst_cos, float((1 - cos).max()))
            print(f"[{key} B{B} S{S} {layout} {dtype} lo={int(lo)}] max abs err {err:.2e}  1-cos {float((1 - cos).max()):.2e}")
            np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
            outs[key] = out; refs[key] = ref
    keys = [k for k in outs if ".e5." in k]
    worst_score = 0.0
    for a in keys:
        for b in keys:
            worst_score = max(worst_score, float(np.abs(outs[a] @ outs[b].T - refs[a] @ refs[b].T).max()))
    print(f"[G10 {wname} {dtype} lo={int(lo)}] worst |q.d - ref| = {worst_score:.2e}, worst abs err {worst_err:.2e}, worst 1-cos {worst_cos:.2e}")
    G10_RESULTS[(wname, dtype, lo)] = (worst_score, worst_err, worst_cos)
    return worst_score, worst_err, worst_cos


# Bars per (weight set, operand type, residual low half): (|q.d - ref|, max abs err, 1 - cos).  The DEFAULT mode (f16 + low half) must meet north_star's
# 1e-3 on the scores on the benign and out3 weight sets (out3 = the outlier level of real checkpoints); out16 is held to 1.5e-3 BY CHOICE (the default measures 1.02e-3
# there: DESIGN.md section 2), out60 and the other modes to what they measure so that a regression shows.
G10_TOL = 1e-3
G10_MODES = [("f16", True), ("f16", False), ("bf16", True), ("bf16", False)]


@pytest.mark.parametrize("wname", ["benign", "out16", "out60", "out3"])      # out3 last: the checkpoint-check test below reuses its cached weights
def test_g10_full_size_long_sequences_big_batch_left_padding_and_outlier_weights(golden, wname):
    """Golden set G10 (generated by importing the reference's E5Encoder / BGEEncoder, tests/golden/make_golden.py g10): the full 24-layer shape at
    S = 256 / 512 (the reference's doc_maxlength default, compute_corpus_embeddings.py:32-33), a 64-sequence batch, left padding — and weights with
    OUTLIER channels (six hidden channels with a large LayerNorm gamma, 8x embedding columns, 10x biases; residual-stream magnitudes against a
    median of ~0.35:  out3 ~80 = the two orders of magnitude of real BERT-family checkpoints, out16 ~400, out60 ~1600).  All four precision modes
    run; the default (f16 operands + residual low half) must meet north_star's bars — scores within 1e-3, cosine to the reference >= 1 - 5e-5 — on the benign
    and out3 sets; on out16 (5 x harsher outliers than real checkpoints) its bar is 1.5e-3, set at what it measures."""
    res = {m: _g10_run(golden, wname, m[0], m[1]) for m in G10_MODES}
    score, err, cos = res[("f16", True)]
    # measured (MI355X, round 3): worst |q.d - ref| of f16+lo / f16 / bf16+lo / bf16 = benign 1.6e-5 / 1.8e-5 / 1.2e-4 / 1.5e-4, out3 1.4e-4 / 7.2e-4 /
    # 1.0e-3 / 5.9e-3, out16 1.0e-3 / 2.0e-3 / 8.9e-3 / 1.9e-2, out60 9.5e-3 / 1.5e-2 / 2.1e-2 / 3.7e-2
    bar = {"benign": G10_TOL, "out3": G10_TOL, "out16": 1.5e-3, "out60": 2e-2}[wname]   # out16 / out60: 6x / 25x harsher than real checkpoints' outliers
    assert score <= bar and cos <= (5e-5 if wname != "out60" else 5e-4) and err <= (3e-3 if wname != "out60" else 3e-2), (wname, score, err, cos)
    if wname == "benign":                                       # every mode is fine on weights without outlier channels
        for m, (sc, er, co) in res.items():
            assert sc <= G10_TOL and co <= (5e-5 if m[0] == "f16" else 1e-4) and er <= 3e-3, (m, sc, er, co)   # bf16: 1 - cos ~ 5e-5 (CLS pooling)
    # the default mode is the most accurate one on every weight set (what justifies paying for it)
    assert score <= min(r[0] for r in res.values()) * 1.5 + 1e-5, res


def test_checkpoint_check_tool_through_the_hip_encoder_on_out3():
    """tools/checkpoint_check.py with the HIP encoder as the tested path, on a full-size BertModel holding G10's out3 weights: the
    four precision modes against the module's own fp32 forward.  The default must stay inside north_star's 1e-3, be the most accurate mode, and the
    8-bit-operand modes must show the outlier channels (bf16 without the low half worst) — the same ordering the golden table has."""
    import importlib.util, os
    import torch
    from transformers import BertConfig, BertModel
    spec = _g10_spec()
    w = _g10_weights(spec, "out3")
    c = spec.CFG
    m = BertModel(BertConfig(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["heads"], intermediate_size=c["FF"], vocab_size=c["vocab"],
                             max_position_embeddings=c["max_pos"], type_vocab_size=2, layer_norm_eps=1e-12), add_pooling_layer=False)
    assert not m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False).missing_keys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("checkpoint_check", os.path.join(repo, "tools", "checkpoint_check.py"))
    cc = importlib.util.module_from_spec(sp); sp.loader.exec_module(cc)
    out = cc.check_model(m, cc.parse(["(out3)", "--n", "24", "--max-length", "128", "--random-tokens"]))
    assert out["tested_path"] == "hip" and 100 < out["outlier_ratio"] < 1000 and out["f16_headroom"] > 100
    err = {(r["operand_dtype"], r["residual_lo"]): r["worst_score_error"] for r in out["modes"]}
    print("checkpoint_check out3 (HIP):", err)
    assert err[("f16", True)] <= 1e-3 and err[("f16", True)] <= min(err.values()) * 1.5 + 1e-5
    assert err[("bf16", False)] > err[("bf16", True)] > err[("f16", True)] and err[("bf16", False)] > err[("f16", False)]


def test_batch_invariance_and_padding_layouts(golden):
    """Same sequences alone / in a batch / left-padded / with interior mask holes give the oracle's answer."""
    g = golden("g1_encoder_tiny.npz")
    cfg = _cfg(g["cfg.t128"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    h = _hip(cfg, w)
    rng = np.random.default_rng(0)
    ids = rng.integers(5, cfg.vocab_size, (6, 40)); mask = np.ones((6, 40), np.int64)
    mask[1, 25:] = 0                    # right padded
    mask[2, :13] = 0                    # left padded (truncate_to_max_sequence's other branch, collators.py:38-44)
    mask[3, 5:9] = 0; mask[3, 30:] = 0  # holes
    mask[4, :] = 0                      # nothing attended -> NaN (encoders.py:56-58)
    mask[5, 1:] = 0                     # a single token
    for pool, fn in ((0, E.e5_encode), (1, E.bge_encode)):
        out = h.forward_np(ids, mask, pool)
        ref = fn(w, ids, mask, cfg.num_attention_heads)
        assert np.isnan(out[4]).all() and np.isnan(ref[4]).all()
        ok = [0, 1, 2, 3, 5]
        _check(out[ok], ref[ok], 4e-3, f"pool{pool}")
        single = h.forward_np(ids[[2]], mask[[2]], pool)
        np.testing.assert_allclose(single[0], out[2], atol=2e-6)       # batch composition does not change a row
    with pytest.raises(Exception) as ei:
        bad = ids.copy(); bad[0, 3] = cfg.vocab_size
        h.forward_np(bad, mask, 0)
    assert "token id" in str(ei.value)


def test_cls_pooling_last_layer_on_cls_rows_only_is_bit_identical(golden, monkeypatch):
    """BGE pooling reads one row per sequence after the last layer: the default path gathers the CLS rows after the last attention and runs the rest of that
    layer on B rows.  It must equal the all-rows computation (KIRAG_AMD_CLS_FULL=1 at creation) bit for bit — ragged, left-padded, holes, all-masked, one
    token, short and long sequences (both attention kernels), many sequences (every projection tiling for B rows) — and mean pooling must not be touched."""
    g = golden("g1_encoder_tiny.npz")
    cfg = _cfg(g["cfg.t256"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    rng = np.random.default_rng(3)
    cases = []
    for B, S in ((6, 40), (3, 200), (300, 33), (70, 64)):
        ids = rng.integers(5, cfg.vocab_size, (B, S)); mask = np.ones((B, S), np.int64)
        lens = rng.integers(1, S + 1, B)
        for b in range(B):
            mask[b, lens[b]:] = 0
        mask[1, :] = 1; mask[1, :S // 3] = 0            # left padded: position 0 masked
        mask[2, 2:5] = 0                                # holes
        if B > 4:
            mask[4, :] = 0                              # nothing attended
        cases.append((ids, mask))
    h_short = _hip(cfg, w)
    monkeypatch.setenv("KIRAG_AMD_CLS_FULL", "1")
    h_full = _hip(cfg, w)
    monkeypatch.delenv("KIRAG_AMD_CLS_FULL")
    for ids, mask in cases:
        a = h_short.forward_np(ids, mask, 1); b = h_full.forward_np(ids, mask, 1)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), ids.shape
        ref = E.bge_encode(w, ids, mask, cfg.num_attention_heads)
        ok = ~np.isnan(ref).any(1)
        _check(a[ok], ref[ok], 4e-3, f"bge {ids.shape}")
        assert np.isnan(a[~ok]).all()
        with pytest.raises(Exception) as ei:
            h_short.last_hidden(*ids.shape)
        assert "CLS" in str(ei.value)
        h_full.last_hidden(*ids.shape)                  # the all-rows encoder still serves the debug read-back
        m = h_short.forward_np(ids, mask, 0); m2 = h_full.forward_np(ids, mask, 0)
        assert np.array_equal(m.view(np.uint32), m2.view(np.uint32))
        h_short.last_hidden(*ids.shape)                 # after a mean-pooled forward every row is final again


def test_both_attention_kernels_give_the_same_bits(golden, monkeypatch):
    """A sequence's embedding must not depend on which attention kernel its batch selects (the embedding cache and batch invariance rely on it): the same
    sequences in a batch of width 100 (register-staged kernel) and padded to width 200 (LDS-DMA ring kernel); and long ragged sequences through the ring
    kernel vs the register-staged one forced by KIRAG_AMD_ATTN_LDS=1 (read per forward)."""
    g = golden("g1_encoder_tiny.npz")
    cfg = _cfg(g["cfg.t256"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    h = _hip(cfg, w)
    rng = np.random.default_rng(7)
    B = 24
    ids = rng.integers(5, cfg.vocab_size, (B, 100)); mask = np.zeros((B, 100), np.int64)
    lens = rng.integers(1, 101, B); lens[0] = 100; lens[1] = 64; lens[2] = 65; lens[3] = 32; lens[4] = 33
    for b in range(B):
        mask[b, :lens[b]] = 1
    wide_ids = np.concatenate([ids, np.zeros((B, 100), np.int64)], 1); wide_mask = np.concatenate([mask, np.zeros((B, 100), np.int64)], 1)
    for pool in (0, 1):
        a = h.forward_np(ids, mask, pool); b_ = h.forward_np(wide_ids, wide_mask, pool)
        assert np.array_equal(a.view(np.uint32), b_.view(np.uint32)), pool
    for S in (129, 200, 300, 512):
        ids = rng.integers(5, cfg.vocab_size, (9, S)); mask = np.zeros((9, S), np.int64)
        lens = rng.integers(1, S + 1, 9); lens[0] = S; lens[1] = 128; lens[2] = 129; lens[3] = 64
        for b in range(9):
            if b % 4 == 3: mask[b, S - lens[b]:] = 1
            else: mask[b, :lens[b]] = 1
        a = h.forward_np(ids, mask, 0)
        monkeypatch.setenv("KIRAG_AMD_ATTN_LDS", "1")
        b_ = h.forward_np(ids, mask, 0)
        monkeypatch.delenv("KIRAG_AMD_ATTN_LDS")
        assert np.array_equal(a.view(np.uint32), b_.view(np.uint32)), S


def test_module_surface_eval_hip_train_torch():
    """E5Encoder/BGEEncoder as nn.Modules: eval -> HIP path, train -> autograd path, CPU eval -> loud failure."""
    import torch
    from transformers import BertConfig
    from kirag_amd.retriever.encoders import BGEEncoder, E5Encoder
    cfg = BertConfig(vocab_size=500, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                     max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    for cls in (E5Encoder, BGEEncoder):
        m = cls(cfg, add_pooling_layer=False)
        m.eval()
        ids = torch.randint(5, 500, (4, 19)); mask = torch.ones(4, 19, dtype=torch.long); mask[1, 7:] = 0
        with pytest.raises(RuntimeError):
            m(ids, mask)                                   # CPU + eval: no fallback
        m = m.cuda()
        out = m(ids.cuda(), mask.cuda())
        assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (4, 128) and not out.requires_grad
        m.train()
        ref = m(ids.cuda(), mask.cuda())                   # PyTorch fp32 autograd path of the same weights
        assert ref.requires_grad
        assert (out - ref.detach()).abs().max().item() <= 4e-3
        # weights changed in place -> the HIP copy must follow
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(1.01)
        m.eval()
        out2 = m(ids.cuda(), mask.cuda())
        m.train(); ref2 = m(ids.cuda(), mask.cuda()).detach(); m.eval()
        assert (out2 - ref2).abs().max().item() <= 4e-3
        assert m.config.hidden_size == 128 and len(list(m.named_parameters())) > 0


@pytest.mark.parametrize("B,S", [(3, 512), (5, 300), (9, 96), (70, 64), (300, 33)])
def test_long_and_many_sequences_vs_oracle(B, S):
    """Sequence lengths up to max_position_embeddings (the reference's doc_maxlength default, compute_corpus_embeddings.py:32-33) and
    batches that span several attention blocks / token tiles, ragged with right and left padding, against the numpy oracle (which is
    pinned to the reference by the goldens).  Covers every heads-per-block variant of the attention kernel (S = 33 / 64 / >= 96) and
    the 256x256, 128x128 and 32x32 projection paths."""
    cfg = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                          max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    w = E.synth_weights(128, 2, 512, 1000, 512, seed=11)
    h = _hip(cfg, w)
    rng = np.random.default_rng(B * 1000 + S)
    ids = rng.integers(5, 1000, (B, S)); mask = np.zeros((B, S), np.int64)
    lens = rng.integers(max(1, S // 3), S + 1, B); lens[0] = S
    for b in range(B):
        if b % 3 == 2:
            mask[b, S - lens[b]:] = 1           # left padded
        else:
            mask[b, :lens[b]] = 1
    for pool, fn in ((0, E.e5_encode), (1, E.bge_encode)):
        out = h.forward_np(ids, mask, pool)
        ref = fn(w, ids, mask, 2)
        _check(out, ref, 4e-3, f"B{B} S{S} pool{pool}")


@pytest.mark.parametrize("tile", ["256", "128", "130", "32"])
def test_every_projection_path_on_every_shape(golden, tile, monkeypatch):
    """The launcher picks the 256x256 ping-pong loop, the 128x128 producer / consumer loop, the 128x128 2-slot streaming loop or the 32x32 skinny
    loop from the token count; force each of them (KIRAG_AMD_PROJ_TILE = 256 / 130 / 128 / 32) on the full-size golden batch and on a ragged
    tiny-config batch."""
    monkeypatch.setenv("KIRAG_AMD_PROJ_TILE", tile)
    g = golden("g2_encoder_large.npz")
    cfg = _cfg(g["cfg"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    h = _hip(cfg, w)
    for key, pool in (("e5.c0", 0), ("e5.c1", 0), ("e5.c2", 0), ("bge.c0", 1)):
        out = h.forward_np(g[key + ".ids"], g[key + ".mask"], pool)
        _check(out, g[key + ".out"], 3e-3, f"tile{tile} {key}")
    del h
    cfg2 = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                           max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    w2 = E.synth_weights(128, 2, 512, 1000, 512, seed=11)
    h2 = _hip(cfg2, w2)
    ids, mask = E.synth_tokens(700, 48, seed=3, ragged=True, vocab_lo=5, vocab_hi=1000, min_len=3)
    _check(h2.forward_np(ids, mask, 0), E.e5_encode(w2, ids, mask, 2), 4e-3, f"tile{tile} tiny ragged")


def test_projection_paths_are_bit_identical(golden, monkeypatch):
    """Every output element is one MFMA accumulator chain over k in increasing order whatever the tile shape, so the main loops must agree
    BIT FOR BIT (the embedding cache and batch-size independence rely on it): full-size golden batch (mean and CLS pooling) and a ragged tiny-config
    batch, through the 256 x 256 ping-pong loop, the two 128 x 128 loops and both skinny tiles (32 x 32, and round 5's 64 x 64)."""
    g = golden("g2_encoder_large.npz")
    cfg = _cfg(g["cfg"])
    w = E.synth_weights(cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings,
                        seed=int(g["weight_seed"]))
    h = _hip(cfg, w)
    cfg2 = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                           max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    h2 = _hip(cfg2, E.synth_weights(128, 2, 512, 1000, 512, seed=11))
    ids2, mask2 = E.synth_tokens(700, 48, seed=3, ragged=True, vocab_lo=5, vocab_hi=1000, min_len=3)
    outs = {}
    for tile in ("256", "128", "130", "32", "64"):
        monkeypatch.setenv("KIRAG_AMD_PROJ_TILE", tile)
        outs[tile] = (h.forward_np(g["e5.c1.ids"], g["e5.c1.mask"], 0), h2.forward_np(ids2, mask2, 0), h.forward_np(g["e5.c1.ids"], g["e5.c1.mask"], 1))
    for tile in outs:
        for i in range(3):
            assert np.array_equal(outs[tile][i].view(np.uint32), outs["256"][i].view(np.uint32)), (tile, i)


def test_small_batch_forwards_replayed_as_hip_graphs_are_bit_identical():
    """KIRAG_AMD_GRAPH=1 (opt-in; measured slower than eager launches on this stack, so off by default): from the second forward of a (B, S, pool) shape
    with at most 4096 tokens on, the ~175 launches of a forward are replayed as one hipGraph.  Replays must equal the eager forwards bit for bit, for
    changing inputs, interleaved shapes, both pooling modes and a workspace re-allocation in between (captured pointers are dropped)."""
    import os
    from types import SimpleNamespace
    import torch
    from oracle import encoder_np as E
    from kirag_amd.retriever.encoders import HipBertForward
    cfg = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                          max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    w = E.synth_weights(128, 2, 512, 1000, 512, seed=5)
    eager = HipBertForward(cfg, 0); eager.load_state(w)
    os.environ["KIRAG_AMD_GRAPH"] = "1"
    try:
        h = HipBertForward(cfg, 0); h.load_state(w)
    finally:
        del os.environ["KIRAG_AMD_GRAPH"]
    shapes = [(1, 32), (2, 256), (8, 128), (1, 32), (4, 64)]
    data = {}
    for rnd in range(4):
        for (B, S) in shapes:
            ids, mask = E.synth_tokens(B, S, seed=100 * rnd + B + S, ragged=True, vocab_lo=5, vocab_hi=1000, min_len=3)
            for pool in (0, 1):
                a = h.forward(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda(), pool)      # eager on round 0, captured on round 1, replayed after
                b = eager.forward(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda(), pool)
                assert torch.equal(a, b), (rnd, B, S, pool)
        if rnd == 1:                                                    # a bigger batch re-allocates the workspace: every captured graph must go
            ids, mask = E.synth_tokens(64, 200, seed=7, ragged=True, vocab_lo=5, vocab_hi=1000, min_len=3)
            assert torch.equal(h.forward(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda(), 0),
                               eager.forward(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda(), 0))
    ids, mask = E.synth_tokens(2, 40, seed=3, ragged=True, vocab_lo=5, vocab_hi=1000, min_len=3)
    ref = E.e5_encode(w, ids, mask, 2)
    for _ in range(3):
        assert np.abs(h.forward_np(ids, mask, 0) - ref).max() < 4e-3    # host-pointer path through the same replay


@pytest.mark.parametrize("B,S", [(1, 32), (5, 300), (9, 96), (64, 40), (65, 64), (300, 33), (513, 128)])
def test_forward_packed_is_bit_identical_to_the_padded_forward(B, S):
    """ABI 9, kr_encoder_forward_packed: the ragged token list of a right-padded batch (int32 attended ids back to back + int32 lengths: what the tokenizer
    processes of the corpus-encode loop ship, kirag_amd/feed.py) gives the SAME BITS as kr_encoder_forward on the padded int64 [B,S] pair
    (dataset/collators.py:59-81 -> encoders.py:67-77 / :106-118), for both pools, the one-block (B <= 64) and the two-kernel packers, every attention
    kernel, sequences of length 0 (mean pool: NaN like an all-zero mask) and S, host pointers and device tensors; and against the numpy oracle."""
    import torch
    cfg = SimpleNamespace(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=1000,
                          max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    w = E.synth_weights(128, 2, 512, 1000, 512, seed=11)
    h = _hip(cfg, w)
    rng = np.random.default_rng(B * 977 + S)
    lens = rng.integers(1, S + 1, B).astype(np.int32); lens[0] = S
    if B > 2:
        lens[B // 2] = 0                                                     # an empty sequence in the middle of the batch
    ids = rng.integers(5, 1000, (B, S)); mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    ids = ids * mask                                                         # [PAD] = 0 at the masked positions, as the tokenizer pads
    rag = np.ascontiguousarray(ids[mask != 0].astype(np.int32))
    for pool, fn in ((0, E.e5_encode), (1, E.bge_encode)):
        padded = h.forward_np(ids, mask, pool)
        out = h.forward_packed(torch.from_numpy(rag), torch.from_numpy(lens), S, pool).cpu().numpy()           # pageable host tensors
        assert np.array_equal(out.view(np.uint32), padded.view(np.uint32)), f"pool {pool}: packed != padded"
        dev = h.forward_packed(torch.from_numpy(rag).cuda(), torch.from_numpy(lens).cuda(), S, pool)           # device tensors, enqueue-only
        pin_ids = torch.from_numpy(np.concatenate([rag, np.full(7, 999_999, np.int32)])).pin_memory()          # pinned slot longer than the batch (total_tokens)
        pin = h.forward_packed(pin_ids, torch.from_numpy(lens).pin_memory(), S, pool, total_tokens=rag.size)
        h.check()
        assert torch.equal(dev.view(torch.int32), pin.view(torch.int32)) and np.array_equal(dev.cpu().numpy().view(np.uint32), padded.view(np.uint32))
        live = lens > 0
        if B > 2:
            assert np.isnan(out[B // 2]).all()                               # nothing attended: NaN for both pools, as for an all-zero mask row
        ref = fn(w, ids, mask, 2)
        _check(out[live], ref[live], 4e-3, f"packed B{B} S{S} pool{pool}")
    # the deferred error channel: lengths that do not add up / a length beyond S / an id outside the vocabulary
    from kirag_amd._lib import KiragAmdError
    for bad_lens, bad_rag, msg in ((np.minimum(lens + (np.arange(B) == 0), S + 1).astype(np.int32), rag, "seq_lens"),
                                   (lens, np.where(np.arange(rag.size) == rag.size - 1, 1000, rag).astype(np.int32), "token id")):
        with pytest.raises(KiragAmdError, match=msg):
            h.forward_packed(torch.from_numpy(bad_rag).cuda(), torch.from_numpy(bad_lens).cuda(), S, 0)
            h.check()
    again = h.forward_packed(torch.from_numpy(rag), torch.from_numpy(lens), S, 0).cpu().numpy()                # the handle is usable again, same bits
    assert np.array_equal(again.view(np.uint32), h.forward_np(ids, mask, 0).view(np.uint32))


def test_module_forward_packed_and_doc_packed_match_the_module_forward():
    """The module surface of the packed path (E5Encoder / BGEEncoder.forward_packed, BaseRetriever.doc_packed) against the module's own eval forward at the
    full e5-large shape with the default (f16 + low half) mode: bit-identical rows; train mode refuses."""
    import torch
    from transformers import BertConfig
    from kirag_amd import bench_support as BS
    from kirag_amd.retriever.encoders import BGEEncoder, E5Encoder
    cfg = BertConfig(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, max_position_embeddings=512)
    torch.manual_seed(0)
    for cls in (E5Encoder, BGEEncoder):
        m = cls(cfg, add_pooling_layer=False).cuda().eval()
        ids, mask = BS.synthetic_tokens(torch.device("cuda:0"), 96, 128, seed=4, ragged=True)
        ref = m(ids, mask)
        lens = mask.sum(1).to(torch.int32)
        rag = ids[mask.bool()].to(torch.int32)
        out = m.forward_packed(rag.cpu(), lens.cpu(), 128)
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
        m._hip.check()
        m.train()
        with pytest.raises(RuntimeError, match="eval"):
            m.forward_packed(rag, lens, 128)
        del m
        torch.cuda.empty_cache()
