"""Drop-in for the reference's ``retriever/e5.py`` (module-level singleton e5 model + ``get_e5_embeddings_for_query`` :64-78 /
``get_e5_embeddings_for_document`` :80-94, used for exemplar ranking by ``knowledge_graph/models.py:1255-1275,1309-1317`` and
``knowledge_graph/kg_generator.py:95-123``).  Same module attributes (``tokenizer``, ``model``, ``device``,
``tokenizer_name_or_path``, ``model_name_or_path``), same signatures, CPU fp32 ``[n, hidden]`` results — the forward runs on the
HIP encoder.  ``set_model`` lets ``retrieve.py``-style callers share the already resident retriever encoder instead of loading a
third copy of e5-large."""
from __future__ import annotations

from typing import Dict, List

import torch
from torch import Tensor

from ..utils import to_device
from .encoders import E5Encoder

tokenizer = None
model = None
tokenizer_name_or_path = 'intfloat/e5-large-v2'
model_name_or_path = 'intfloat/e5-large-v2'
device = torch.device("cuda")


def get_tokenizer():
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(tokenizer_name_or_path)


def get_model():
    print(f"loading E5 checkpoint from {model_name_or_path} ... ")
    m = E5Encoder.from_pretrained(model_name_or_path)
    m.to(device)
    m.eval()
    return m


def set_model(encoder, tok=None) -> None:
    """Share an already loaded E5Encoder (and tokenizer) with this module."""
    global model, tokenizer
    model = encoder
    if tok is not None:
        tokenizer = tok


def tokenizer_encode(texts: List[str], max_length: int):
    global tokenizer
    tokenizer = get_tokenizer() if tokenizer is None else tokenizer
    batch = tokenizer(texts, max_length=max_length, padding=True, truncation=True, return_tensors='pt')
    return {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"]}


def model_encode(inputs: Dict[str, Tensor]) -> Tensor:
    """e5.py:51-61 — forward -> average_pool -> F.normalize; all three are fused in the HIP encoder (eval mode)."""
    global model
    model = get_model() if model is None else model
    dev = next(model.parameters()).device
    inputs = to_device(inputs, dev)
    with torch.no_grad():
        return model(**inputs)


def _embed(texts: List[str], max_length: int, batch_size: int) -> Tensor:
    # rows are independent of batch composition on the HIP path: one launch for the whole (short) list
    step = max(batch_size, 256)
    # every chunk is only ENQUEUED; the one device-to-host copy at the end is the only synchronisation (the reference copies per batch of 4)
    outs = [model_encode(tokenizer_encode(texts[i:i + step], max_length=max_length)).detach() for i in range(0, len(texts), step)]
    out = torch.cat(outs, dim=0).cpu()
    hip = getattr(model, "_hip", None)
    if hip is not None:
        hip.check()          # deferred input errors of the forwards above (token id outside the vocabulary, token_type_ids != 0, non-finite activations)
    return out               # surface here, behind the copy that already waited for the device — never as plausible-looking vectors


def get_e5_embeddings_for_query(query_list: List[str], max_length: int = 128, batch_size: int = 4) -> Tensor:
    return _embed(["query: " + q for q in query_list], max_length, batch_size)


def get_e5_embeddings_for_document(doc_list: List[str], max_length: int = 256, batch_size: int = 4) -> Tensor:
    return _embed(["passage: " + d for d in doc_list], max_length, batch_size)
