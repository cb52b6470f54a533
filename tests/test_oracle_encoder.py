"""Pins oracle/encoder_np.py against vectors produced by the reference's own encoders
(tests/golden/make_golden.py, imported from /root/reference in the build container)."""
import numpy as np
import pytest

from oracle import encoder_np as E


def _cfg(arr):
    H, L, heads, FF, vocab, max_pos = [int(v) for v in arr]
    return dict(H=H, L=L, heads=heads, FF=FF, vocab=vocab, max_pos=max_pos)


def test_g1_tiny_encoders_match_reference(golden):
    g = golden("g1_encoder_tiny.npz")
    seed = int(g["weight_seed"])
    n = 0
    for name in g["cfg_names"]:
        c = _cfg(g[f"cfg.{name}"])
        w = E.synth_weights(c["H"], c["L"], c["FF"], c["vocab"], c["max_pos"], seed=seed)
        for tag, fn in (("e5", E.e5_encode), ("bge", E.bge_encode)):
            for ci in range(7):
                key = f"{name}.{tag}.c{ci}"
                ids, mask, ref = g[key + ".ids"], g[key + ".mask"], g[key + ".out"]
                out = fn(w, ids, mask, c["heads"])
                assert out.shape == ref.shape
                np.testing.assert_allclose(out, ref, atol=2e-5, rtol=0)   # fp32 vs fp32, different BLAS order
                np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
                n += 1
    assert n == 3 * 2 * 7


def test_g1_hidden_states(golden):
    g = golden("g1_encoder_tiny.npz")
    seed = int(g["weight_seed"])
    for name in g["cfg_names"]:
        c = _cfg(g[f"cfg.{name}"])
        w = E.synth_weights(c["H"], c["L"], c["FF"], c["vocab"], c["max_pos"], seed=seed)
        for ci in (1, 3):
            key = f"{name}.e5.c{ci}"
            ids, mask, hid = g[key + ".ids"], g[key + ".mask"], g[key + ".hidden"]
            _, hs = E.bert_forward(w, ids, mask, c["heads"], return_all=True)
            assert len(hs) == hid.shape[0] == c["L"] + 1
            keep = mask.astype(bool)
            for l in range(len(hs)):   # padded query rows are don't-care (never pooled)
                np.testing.assert_allclose(hs[l][keep], hid[l][keep], atol=3e-5, rtol=0)


def test_g3_average_pool_and_nan_row(golden):
    g = golden("g3_average_pool.npz")
    out = E.average_pool(g["lh"], g["mask"])
    assert np.isnan(out[2]).all() and np.isnan(g["out"][2]).all()
    np.testing.assert_allclose(out[[0, 1, 3]], g["out"][[0, 1, 3]], atol=1e-6)


def test_g4_logits_and_score(golden):
    g = golden("g4_g8_retriever.npz")
    q1, d1, q2, d2, d3 = (g[f"g4.{k}"] for k in ("q1", "d1", "q2", "d2", "d3"))
    np.testing.assert_allclose(E.compute_logits(q1, d1), g["g4.l11"], rtol=1e-5)
    np.testing.assert_allclose(E.compute_logits(q1, d2), g["g4.l12"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(E.compute_logits(q2, d3), g["g4.l23"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(E.compute_logits(q2, d2), g["g4.l22"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(E.score(q2, d2, 0.01), g["g4.s22_t001"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(E.score(q2, d2, "sqrt"), g["g4.s22_sqrt"], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError) as ei:
        E.compute_logits(d3, d3)
    assert "Invalid embedding shape" in str(ei.value) and "Invalid embedding shape" in str(g["g4.err"])


def test_g5_g6_g8_through_oracle(golden):
    g = golden("g4_g8_retriever.npz")
    c = _cfg(g["cfg"])
    w = E.synth_weights(c["H"], c["L"], c["FF"], c["vocab"], c["max_pos"], seed=int(g["weight_seed"]))
    ids, mask = g["g5.ids"], g["g5.mask"]
    out = E.e5_encode(w, ids.reshape(-1, ids.shape[-1]), mask.reshape(-1, ids.shape[-1]), c["heads"]).reshape(2, 3, -1)
    np.testing.assert_allclose(out, g["g5.out"], atol=2e-5)
    # G6: in-batch scores = q d^T / 0.01, CE over rows with labels 0..2
    q = E.e5_encode(w, g["g7.e5.q.ids"], g["g7.e5.q.mask"], c["heads"])
    d = E.e5_encode(w, g["g7.e5.d.ids"], g["g7.e5.d.mask"], c["heads"])
    np.testing.assert_allclose(q, g["g6.q"], atol=2e-5)
    np.testing.assert_allclose(d, g["g6.d"], atol=2e-5)
    sc = E.score(q, d, 0.01)
    np.testing.assert_allclose(sc, g["g6.scores"], atol=5e-3)
    lse = np.log(np.exp(sc - sc.max(1, keepdims=True)).sum(1)) + sc.max(1)
    loss = float(np.mean(lse - sc[np.arange(3), np.arange(3)]))
    assert abs(loss - float(g["g6.loss"])) < 5e-3


@pytest.mark.slow
def test_g2_large_shape_matches_reference(golden):
    """Full e5-large shape (24 x 1024): one ragged [8,128] batch through the numpy oracle (~20 s)."""
    g = golden("g2_encoder_large.npz")
    c = _cfg(g["cfg"])
    w = E.synth_weights(c["H"], c["L"], c["FF"], c["vocab"], c["max_pos"], seed=int(g["weight_seed"]))
    for key, fn in (("e5.c1", E.e5_encode), ("bge.c0", E.bge_encode)):
        out = fn(w, g[key + ".ids"], g[key + ".mask"], c["heads"])
        np.testing.assert_allclose(out, g[key + ".out"], atol=5e-5)


def test_checkpoint_check_tool_on_an_outlier_checkpoint(tmp_path):
    """tools/checkpoint_check.py (a precision self-check a user WITH a real checkpoint can run) end to end on the CPU: a small BERT
    with the outlier-channel recipe of golden set G10 saved as an HF directory, activation statistics from the module's own fp32 forward, the four precision
    modes through the torch emulation of the HIP encoder's rounding points.  Must reproduce G10's ordering — f16 + low half best, bf16 worst, the low half
    helping either operand type — and report an outlier ratio far above a benign model's."""
    import importlib.util
    import json
    import os
    import torch
    from transformers import BertConfig, BertModel
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("checkpoint_check", os.path.join(repo, "tools", "checkpoint_check.py"))
    cc = importlib.util.module_from_spec(spec); spec.loader.exec_module(cc)
    H, L, FF, vocab = 256, 6, 1024, 2000
    res = {}
    for name, w in (("benign", E.synth_weights(H, L, FF, vocab, 512, seed=3)), ("outlier", E.synth_weights_outlier(H, L, FF, vocab, 512, seed=7, gamma_lo=1.5, gamma_hi=3.0))):
        m = BertModel(BertConfig(hidden_size=H, num_hidden_layers=L, num_attention_heads=H // 64, intermediate_size=FF, vocab_size=vocab, max_position_embeddings=512,
                                 type_vocab_size=2, layer_norm_eps=1e-12), add_pooling_layer=False)
        assert not m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False).missing_keys
        d = tmp_path / name
        m.save_pretrained(str(d))
        out = cc.check(str(d), cc.parse([str(d), "--n", "12", "--max-length", "96", "--emulate", "--json", str(tmp_path / (name + ".json"))]))
        assert json.load(open(tmp_path / (name + ".json")))["recommendation"] == out["recommendation"]
        res[name] = out
    err = {(r["operand_dtype"], r["residual_lo"]): r["worst_score_error"] for r in res["outlier"]["modes"]}
    assert err[("f16", True)] < err[("f16", False)] < err[("bf16", False)] and err[("f16", True)] < err[("bf16", True)] < err[("bf16", False)], err
    assert err[("f16", True)] < 1e-3 and res["outlier"]["outlier_ratio"] > 5 * res["benign"]["outlier_ratio"] and res["outlier"]["f16_headroom"] > 100
    assert len(res["outlier"]["layers"]) == L and all(r["finite"] for r in res["outlier"]["modes"])
