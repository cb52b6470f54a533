"""Synthetic-weight encoder construction for bench.py / smoke (BERT-large shape, N(0,0.02) weights,
LayerNorm gamma=1 beta=0; throughput does not depend on the weight values).  Weights are generated on the device."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .retriever.encoders import HipBertForward

E5_LARGE = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, vocab_size=30522,
                max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")


def synthetic_state(cfg: SimpleNamespace, device, seed: int = 0):
    g = torch.Generator(device=device); g.manual_seed(seed)
    H, FF = cfg.hidden_size, cfg.intermediate_size

    def n(*shape):
        return 0.02 * torch.randn(*shape, generator=g, device=device)

    yield "embeddings.word_embeddings.weight", n(cfg.vocab_size, H)
    yield "embeddings.position_embeddings.weight", n(cfg.max_position_embeddings, H)
    yield "embeddings.token_type_embeddings.weight", n(cfg.type_vocab_size, H)
    yield "embeddings.LayerNorm.weight", torch.ones(H, device=device)
    yield "embeddings.LayerNorm.bias", torch.zeros(H, device=device)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        for nm in ("query", "key", "value"):
            yield p + f"attention.self.{nm}.weight", n(H, H)
            yield p + f"attention.self.{nm}.bias", torch.zeros(H, device=device)
        yield p + "attention.output.dense.weight", n(H, H)
        yield p + "attention.output.dense.bias", torch.zeros(H, device=device)
        yield p + "attention.output.LayerNorm.weight", torch.ones(H, device=device)
        yield p + "attention.output.LayerNorm.bias", torch.zeros(H, device=device)
        yield p + "intermediate.dense.weight", n(FF, H)
        yield p + "intermediate.dense.bias", torch.zeros(FF, device=device)
        yield p + "output.dense.weight", n(H, FF)
        yield p + "output.dense.bias", torch.zeros(H, device=device)
        yield p + "output.LayerNorm.weight", torch.ones(H, device=device)
        yield p + "output.LayerNorm.bias", torch.zeros(H, device=device)


def make_hip_encoder(device, cfg_dict=None, seed: int = 0) -> HipBertForward:
    cfg = SimpleNamespace(**(cfg_dict or E5_LARGE))
    idx = device.index if device.index is not None else torch.cuda.current_device()
    enc = HipBertForward(cfg, idx)
    enc.load_state(dict(synthetic_state(cfg, device, seed)))
    enc.cfg = cfg
    return enc


def synthetic_tokens(device, n: int, S: int, seed: int, ragged: bool = False):
    """ids uniform in [1000, 30000), ids[:,0]=101, last real token 102, right padded."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    ids = torch.randint(1000, 30000, (n, S), generator=g, device=device, dtype=torch.int64)
    if ragged:
        lens = torch.clamp(torch.round(torch.randn(n, generator=g, device=device) * 0.2 * S + 0.86 * S), min(16, S), S).to(torch.int64)
    else:
        lens = torch.full((n,), S, device=device, dtype=torch.int64)
    mask = (torch.arange(S, device=device)[None, :] < lens[:, None]).to(torch.int64)
    ids[:, 0] = 101
    ids[torch.arange(n, device=device), lens - 1] = 102
    return ids * mask, mask


def encoder_flops(cfg, lens) -> float:
    """Algorithmic FLOPs: per sequence of length s: L * s * (24 H^2 + 4 s H)."""
    H, L = cfg.hidden_size, cfg.num_hidden_layers
    lens = lens.double()
    return float((L * lens * (24.0 * H * H + 4.0 * lens * H)).sum().item())
