"""Torch emulation of the HIP encoder's ROUNDING POINTS (16-bit MFMA operands, fp32 accumulation, 16-bit stored activations, optional (hi, lo) residual
stream) — shared by tools/precision_probe.py (design tool, golden set G10) and tools/checkpoint_check.py (a user's own checkpoint).  Pure torch, CPU or
GPU; W is an HF BertModel state dict (name -> tensor)."""
import math

import torch


def rnd(t, mode):
    if mode == "bf16": return t.to(torch.bfloat16).float()
    if mode == "f16": return t.to(torch.float16).float()
    return t


def forward(W, ids, mask, heads, op, resid_lo, y16=True, pool="mean", site=None):
    """op: 'bf16' | 'f16' | 'f32' = precision of every MFMA operand and of every stored activation.  site: optional {name: precision} overriding op for
    single rounding points — 'wqkv' 'w1' 'wo' 'w2' (weights), 'ctx' 'h' (the activations only the out-proj / FF2 GEMMs read), 'y' (dense outputs) —
    e.g. {'h': 'bf16', 'w2': 'bf16'} = the FF2 GEMM on bf16 operands inside an f16 encoder (round 4: which GEMMs need the 11 bits?)."""
    site = site or {}
    def sop(name): return site.get(name, op)
    B, S = ids.shape
    H = W["embeddings.word_embeddings.weight"].shape[1]; dh = H // heads
    def ln(x, g, b): return torch.nn.functional.layer_norm(x, (H,), g, b, 1e-12)
    def stream(x, which):     # what the next GEMM reads (hi) and what the residual add sees (hi [+ lo]); which: 0 embedding, 1 after LN1, 2 after LN2
        hi = rnd(x, op)
        use = resid_lo in (1, True) or (resid_lo == 2 and which in (0, 2)) or (resid_lo == 3 and which == 1)
        if resid_lo == 4:        # low half as 8 bits of hi's ulp (19 significand bits with f16)
            mant = 10 if op == "f16" else 7
            ulp = torch.exp2(torch.floor(torch.log2(hi.abs().clamp_min(1e-30))) - mant)
            lo = torch.clamp(torch.round((x - hi) / ulp * 256), -128, 127) / 256 * ulp
            return hi, hi + lo
        return hi, (hi + rnd(x - hi, op) if use else hi)
    x = W["embeddings.word_embeddings.weight"][ids] + W["embeddings.position_embeddings.weight"][:S][None] + W["embeddings.token_type_embeddings.weight"][0]
    x = ln(x, W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"])
    xh, xr = stream(x, 0)
    keep = mask.bool()[:, None, None, :]
    L = 0
    while f"encoder.layer.{L}.attention.self.query.weight" in W: L += 1
    for l in range(L):
        p = f"encoder.layer.{l}."
        def lin(t, name): return t @ rnd(W[p + name + ".weight"], sop("w1" if name.startswith("intermediate") else "wqkv")).T + W[p + name + ".bias"]
        q = rnd(lin(xh, "attention.self.query") / math.sqrt(dh), op).view(B, S, heads, dh).transpose(1, 2)
        k = rnd(lin(xh, "attention.self.key"), op).view(B, S, heads, dh).transpose(1, 2)
        v = rnd(lin(xh, "attention.self.value"), op).view(B, S, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)).masked_fill(~keep, float("-inf"))
        pr = torch.softmax(s, -1)
        m = s.max(-1, keepdim=True).values
        e = rnd(torch.exp(s - m), op)                       # P is fed to the MFMA as 16-bit, the row sum is kept in fp32
        ctx = rnd(((e @ v) / torch.exp(s - m).sum(-1, keepdim=True)), sop("ctx")).transpose(1, 2).reshape(B, S, H)
        y = ctx @ rnd(W[p + "attention.output.dense.weight"], sop("wo")).T
        y = rnd(y, sop("y")) if y16 else y
        x = ln(y + W[p + "attention.output.dense.bias"] + xr, W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"])
        xh, xr = stream(x, 1)
        h = rnd(torch.nn.functional.gelu(lin(xh, "intermediate.dense")), sop("h"))
        y = h @ rnd(W[p + "output.dense.weight"], sop("w2")).T
        y = rnd(y, sop("y")) if y16 else y
        x = ln(y + W[p + "output.dense.bias"] + xr, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"])
        if l + 1 < L: xh, xr = stream(x, 2)
    if pool == "mean":
        mm = mask[..., None].float()
        emb = (x * mm).sum(1) / mm.sum(1)
    else:
        emb = x[:, 0]
    return torch.nn.functional.normalize(emb, dim=1)
