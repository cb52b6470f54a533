"""CPU timing stand-in for the reference's encoder loop (TEST INFRASTRUCTURE — see oracle/__init__.py).

The reference's encoders are ``transformers.BertModel`` subclasses run in fp32 (``retriever/encoders.py:61-77``; query loop
``retriever/retrievers.py:194-212``: collate -> forward -> ``.detach().cpu()`` per batch).  This module rebuilds exactly that
— HF ``BertModel`` at the e5-large-v2 shape with random weights, masked mean pool, L2 normalise, batches of
``per_gpu_batch_size`` = 8 (``compute_corpus_embeddings.py:43``, ``retrieve.py:116``) — for ``bench.py``'s ``cpu_baseline``."""
from __future__ import annotations

import time

import torch


def build_cpu_e5_large():
    from transformers import BertConfig, BertModel
    cfg = BertConfig(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                     max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)
    torch.manual_seed(0)
    return BertModel(cfg, add_pooling_layer=False).eval()


def encode(model, ids, mask):
    """retriever/encoders.py:67-77."""
    lh = model(input_ids=ids, attention_mask=mask, return_dict=True).last_hidden_state
    lh = lh.masked_fill(~mask[..., None].bool(), 0.0)
    emb = lh.sum(dim=1) / mask.sum(dim=1)[..., None]
    return torch.nn.functional.normalize(emb, p=2, dim=1)


def time_encode(n_seq: int, S: int, batch: int = 8, repeats: int = 5):
    """-> (sequences per second, median seconds of one pass, all pass times) for n_seq sequences of S tokens in batches of `batch`.
    BASELINE.md section 3 protocol: one warm-up pass, then the MEDIAN of `repeats` timed passes over the same sample.  (Reference: no no_grad
    in cal_doc_embeddings, but autograd bookkeeping is not the arithmetic being compared: timed under no_grad, i.e. in the CPU's favour.)"""
    model = build_cpu_e5_large()
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1000, 30000, (n_seq, S), generator=g); mask = torch.ones(n_seq, S, dtype=torch.long)
    times = []
    with torch.no_grad():
        for rep in range(repeats + 1):
            t0 = time.perf_counter()
            for s in range(0, n_seq, batch):
                encode(model, ids[s:s + batch], mask[s:s + batch]).detach().cpu()
            if rep:                                          # pass 0 = warm-up
                times.append(time.perf_counter() - t0)
    times.sort()
    dt = times[len(times) // 2]
    return n_seq / dt, dt, times
