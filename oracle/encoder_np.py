"""NumPy restatement of the reference's sentence encoders (TEST ORACLE — see oracle/__init__.py).

Follows, line by line:
  * ``retriever/encoders.py:56-58``  ``average_pool``  (= ``retriever/e5.py:46-48``)
  * ``retriever/encoders.py:61-77``  ``E5Encoder.forward``  (BertModel -> mean-pool -> L2 normalise)
  * ``retriever/encoders.py:100-118`` ``BGEEncoder.forward`` (BertModel -> [:,0] -> L2 normalise)
  * HF ``BertModel.forward`` semantics reached from those (third-party, transformers==4.44.2 pin,
    ``requirements.txt:9``): embeddings = word + position(0..S-1) + token_type(0) -> LayerNorm;
    L x { Q,K,V = xW^T+b ; softmax(QK^T/sqrt(d_h) + key_mask) V ; dense + residual -> LayerNorm ;
    dense -> erf-GELU ; dense + residual -> LayerNorm } ; pooler unused.

Weights are given as a dict keyed by the HF ``state_dict`` names of ``BertModel`` (what
``E5Encoder.from_pretrained`` loads), values ``np.ndarray``.  All arithmetic is done in ``dtype``
(float32 like the reference, or float64 to get a high-precision truth).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

try:  # scipy is available in the image; fall back to math.erf vectorised
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    # torch.nn.LayerNorm: biased variance over the last dim
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def gelu_erf(x: np.ndarray) -> np.ndarray:
    # HF ACT2FN["gelu"] = exact erf GELU
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def bert_forward(
    weights: Dict[str, np.ndarray],
    input_ids: np.ndarray,
    attention_mask: np.ndarray,
    num_heads: int,
    eps: float = 1e-12,
    dtype=np.float32,
    token_type_ids: Optional[np.ndarray] = None,
    return_all: bool = False,
):
    """BertModel last_hidden_state.  input_ids/attention_mask: int [B,S] (right- or left-padded)."""
    W = {k: np.asarray(v, dtype=dtype) for k, v in weights.items() if "position_ids" not in k}
    ids = np.asarray(input_ids)
    mask = np.asarray(attention_mask)
    B, S = ids.shape
    H = W["embeddings.word_embeddings.weight"].shape[1]
    dh = H // num_heads
    tt = np.zeros_like(ids) if token_type_ids is None else np.asarray(token_type_ids)

    x = (
        W["embeddings.word_embeddings.weight"][ids]
        + W["embeddings.position_embeddings.weight"][np.arange(S)][None]
        + W["embeddings.token_type_embeddings.weight"][tt]
    )
    x = layer_norm(x, W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], eps)
    hidden: List[np.ndarray] = [x]

    key_keep = mask.astype(bool)[:, None, None, :]  # [B,1,1,S]
    L = 0
    while f"encoder.layer.{L}.attention.self.query.weight" in W:
        L += 1
    for l in range(L):
        p = f"encoder.layer.{l}."

        def lin(t, name):
            return t @ W[p + name + ".weight"].T + W[p + name + ".bias"]

        q = lin(x, "attention.self.query").reshape(B, S, num_heads, dh).transpose(0, 2, 1, 3)
        k = lin(x, "attention.self.key").reshape(B, S, num_heads, dh).transpose(0, 2, 1, 3)
        v = lin(x, "attention.self.value").reshape(B, S, num_heads, dh).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) / dtype(math.sqrt(dh))
        # additive key mask: masked keys get probability exactly 0 (finfo.min / -inf in HF);
        # a row whose keys are ALL masked is NaN under sdpa -inf masking; the pooled output of such a
        # sequence is NaN in the reference regardless (average_pool divides by mask.sum()==0).
        s = np.where(key_keep, s, -np.inf)
        with np.errstate(invalid="ignore"):
            s = s - s.max(axis=-1, keepdims=True)
            e = np.exp(s)
            pr = e / e.sum(axis=-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        a = lin(ctx, "attention.output.dense") + x
        x = layer_norm(a, W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"], eps)
        h = gelu_erf(lin(x, "intermediate.dense"))
        o = lin(h, "output.dense") + x
        x = layer_norm(o, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], eps)
        hidden.append(x)
    return (x, hidden) if return_all else x


def average_pool(last_hidden: np.ndarray, attention_mask: np.ndarray) -> np.ndarray:
    """retriever/encoders.py:56-58 — masked_fill(~mask, 0).sum(1) / mask.sum(1); all-masked row -> NaN."""
    m = np.asarray(attention_mask).astype(bool)
    lh = np.where(m[..., None], last_hidden, 0.0).astype(last_hidden.dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        return lh.sum(axis=1) / np.asarray(attention_mask).sum(axis=1)[..., None].astype(last_hidden.dtype)


def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """torch.nn.functional.normalize(p=2, dim=1): x / max(||x||, eps)."""
    n = np.sqrt((x * x).sum(axis=1, keepdims=True))
    return x / np.maximum(n, eps)


def e5_encode(weights, input_ids, attention_mask, num_heads, eps=1e-12, dtype=np.float32):
    """retriever/encoders.py:67-77."""
    lh = bert_forward(weights, input_ids, attention_mask, num_heads, eps, dtype)
    return l2_normalize(average_pool(lh, attention_mask))


def bge_encode(weights, input_ids, attention_mask, num_heads, eps=1e-12, dtype=np.float32):
    """retriever/encoders.py:106-118."""
    lh = bert_forward(weights, input_ids, attention_mask, num_heads, eps, dtype)
    return l2_normalize(lh[:, 0])


def compute_logits(q: np.ndarray, d: np.ndarray) -> np.ndarray:
    """retriever/retrievers.py:71-84 — the four einsum rank cases; anything else raises ValueError."""
    if q.ndim == 1 and d.ndim == 1:
        return np.einsum("d,d->", q, d)
    if q.ndim == 1 and d.ndim == 2:
        return np.einsum("d,md->m", q, d)
    if q.ndim == 2 and d.ndim == 3:
        assert len(q) == len(d)
        return np.einsum("nd,nmd->nm", q, d)
    if q.ndim == 2 and d.ndim == 2:
        return np.einsum("nd,md->nm", q, d)
    raise ValueError(f"Invalid embedding shape! query_embeddings: {q.shape}, doc_embeddings: {d.shape}.")


def score(q: np.ndarray, d: np.ndarray, temperature) -> np.ndarray:
    """retriever/retrievers.py:86-91."""
    if temperature == "sqrt":
        return compute_logits(q, d) / np.sqrt(q.shape[-1])
    return compute_logits(q, d) / temperature


# ---------------------------------------------------------------------------------------------
# synthetic weights / inputs of SURVEY.md §8(d) — seeded recipe, regenerated wherever needed
# ---------------------------------------------------------------------------------------------
def bert_param_shapes(H: int, L: int, FF: int, vocab: int, max_pos: int = 512, type_vocab: int = 2):
    shapes = {
        "embeddings.word_embeddings.weight": (vocab, H),
        "embeddings.position_embeddings.weight": (max_pos, H),
        "embeddings.token_type_embeddings.weight": (type_vocab, H),
        "embeddings.LayerNorm.weight": (H,),
        "embeddings.LayerNorm.bias": (H,),
    }
    for l in range(L):
        p = f"encoder.layer.{l}."
        for n in ("query", "key", "value"):
            shapes[p + f"attention.self.{n}.weight"] = (H, H)
            shapes[p + f"attention.self.{n}.bias"] = (H,)
        shapes[p + "attention.output.dense.weight"] = (H, H)
        shapes[p + "attention.output.dense.bias"] = (H,)
        shapes[p + "attention.output.LayerNorm.weight"] = (H,)
        shapes[p + "attention.output.LayerNorm.bias"] = (H,)
        shapes[p + "intermediate.dense.weight"] = (FF, H)
        shapes[p + "intermediate.dense.bias"] = (FF,)
        shapes[p + "output.dense.weight"] = (H, FF)
        shapes[p + "output.dense.bias"] = (H,)
        shapes[p + "output.LayerNorm.weight"] = (H,)
        shapes[p + "output.LayerNorm.bias"] = (H,)
    return shapes


def synth_weights(H=1024, L=24, FF=4096, vocab=30522, max_pos=512, seed=0, nontrivial=True):
    """PCG64-seeded BERT-shaped weights.  Linear/embedding ~ N(0, 0.02).  With ``nontrivial`` the
    biases are N(0,0.02) and LayerNorm gamma ~ 1+N(0,0.05), beta ~ N(0,0.02) so that every parameter
    influences the output (a gamma=1/beta=0/bias=0 model would hide indexing mistakes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in bert_param_shapes(H, L, FF, vocab, max_pos).items():
        if name.endswith("LayerNorm.weight"):
            w = 1.0 + (0.05 * rng.standard_normal(shp) if nontrivial else 0.0)
        elif name.endswith("LayerNorm.bias") or name.endswith(".bias"):
            w = 0.02 * rng.standard_normal(shp) if nontrivial else np.zeros(shp)
        else:
            w = 0.02 * rng.standard_normal(shp)
        out[name] = np.asarray(w, dtype=np.float32)
    return out


def synth_weights_outlier(H=1024, L=24, FF=4096, vocab=30522, max_pos=512, seed=7, n_outlier=6, gamma_lo=30.0, gamma_hi=60.0):
    """``synth_weights`` plus the statistics real BERT-family checkpoints show and N(0, 0.02) weights do not: a handful of
    "outlier" hidden channels, the same in every layer, whose LayerNorm gamma is 30-60x the others (either sign) with an O(1) beta,
    word / position embedding columns 8x larger in those channels, and biases an order of magnitude larger than the benign recipe's
    (N(0, 0.2) for the linear layers).  The LayerNorm outputs then carry +-30...+-150 in those channels against O(1) elsewhere, which is the
    regime where 16-bit GEMM operands and a 16-bit residual stream lose the most.  Seeded; regenerated on both sides, never stored."""
    out = synth_weights(H, L, FF, vocab, max_pos, seed=seed, nontrivial=True)
    rng = np.random.Generator(np.random.PCG64(seed + 10_000))
    ch = np.sort(rng.choice(H, size=n_outlier, replace=False))
    for name in list(out):
        w = out[name]
        if name.endswith("LayerNorm.weight"):
            w[ch] = rng.uniform(gamma_lo, gamma_hi, size=n_outlier) * rng.choice([-1.0, 1.0], size=n_outlier)
        elif name.endswith("LayerNorm.bias"):
            w[ch] = rng.standard_normal(n_outlier)
        elif name.endswith(".bias"):
            w *= 10.0
        elif name in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight"):
            w[:, ch] *= 8.0
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def synth_tokens(n: int, S: int, seed: int, ragged: bool = False, vocab_lo=1000, vocab_hi=30000, min_len=16):
    """ids uniform in [vocab_lo, vocab_hi), ids[:,0]=101, last real token 102,
    right-padded with 0; lengths fixed S, or ragged clip(N(0.86 S, 0.2 S), min_len, S)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = rng.integers(vocab_lo, vocab_hi, size=(n, S), dtype=np.int64)
    if ragged:
        lens = np.clip(np.rint(rng.normal(0.86 * S, 0.2 * S, size=n)), min(min_len, S), S).astype(np.int64)
    else:
        lens = np.full(n, S, dtype=np.int64)
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    ids[:, 0] = 101
    ids[np.arange(n), lens - 1] = 102
    ids = ids * mask
    return ids, mask
