"""Drop-in for the reference's ``retriever/retrievers.py``: the plugin registry (``RETRIEVER_MAP`` / ``load_retriever``
:20-29) and the retriever classes (``BaseRetriever`` :32-128, ``InBatchRetriever`` :131-150, ``DenseRetriever`` :153-291)
over the MI355X encoders (``kirag_amd.retriever.encoders``) and index (``kirag_amd.retriever.index``).

Behavioural contract kept: constructor signatures, ``query``/``doc`` accepting ``input_ids`` of rank >= 2, the four
``compute_logits`` rank cases and its ``ValueError``, ``score`` with a numeric temperature or ``"sqrt"``, tuple outputs
unwrapped to ``[0]``, ``DenseRetriever`` result structure (``"score"`` is ``float`` with a corpus, ``np.float32`` without;
ids are ``str``), the ``max_length`` kwarg override, and the asserts on empty input / missing indexer.

What is different underneath: encoding happens on the HIP path and — because a row's embedding does not depend on
what else is in the batch (packed tokens) — ``DenseRetriever`` encodes ``encode_batch_size`` texts per launch instead of
``batch_size`` (4 in the reference) and keeps the embeddings on the GPU until they are needed; the search runs in HBM.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, List, Optional, Union

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
from torch import Tensor

from ..utils import get_global_embeddings_for_inbatchtraining, get_global_labels_for_inbatchtraining, to_device
from .encoders import BGEEncoder, E5Encoder
from .index import Indexer

RETRIEVER_MAP = {
    "E5Retriever": E5Encoder,
    "BGERetriever": BGEEncoder,
}


def load_retriever(retriever_name, model_name_or_path, **kwargs):
    if retriever_name not in RETRIEVER_MAP:
        raise KeyError(f"{retriever_name} is not implemented! Current available retrievers: {list(RETRIEVER_MAP.keys())}")
    print(f"loading {retriever_name} model from {model_name_or_path} ...")
    return RETRIEVER_MAP[retriever_name].from_pretrained(model_name_or_path, **kwargs)


class BaseRetriever(nn.Module):

    def __init__(self, retriever_name, model_name_or_path, retriever_kwargs={}, temperature=1.0, norm_query=False, norm_doc=False,
                 local_rank=-1, **kwargs):
        super().__init__()
        self.encoder = load_retriever(retriever_name, model_name_or_path, **retriever_kwargs, **kwargs)
        self.retriever_name = retriever_name
        self.model_name_or_path = model_name_or_path
        self.retriever_kwargs = retriever_kwargs
        self.norm_query, self.norm_doc = norm_query, norm_doc
        self.local_rank = local_rank
        self.world_size = dist.get_world_size() if self.local_rank >= 0 else 1
        self.temperature = temperature
        self.kwargs = kwargs

    @property
    def device(self):
        for _, p in self.named_parameters():
            return p.device

    @property
    def hidden_size(self):
        return self.encoder.config.hidden_size

    def compute_logits(self, query_embeddings, doc_embeddings, **kwargs):
        nq, nd = query_embeddings.dim(), doc_embeddings.dim()
        if nq == 1 and nd == 1:
            return torch.einsum("d,d->", query_embeddings, doc_embeddings)
        if nq == 1 and nd == 2:
            return torch.einsum("d,md->m", query_embeddings, doc_embeddings)
        if nq == 2 and nd == 3:
            assert len(query_embeddings) == len(doc_embeddings)
            return torch.einsum("nd,nmd->nm", query_embeddings, doc_embeddings)
        if nq == 2 and nd == 2:
            return torch.einsum("nd,md->nm", query_embeddings, doc_embeddings)
        raise ValueError(f"Invalid embedding shape! query_embeddings: {query_embeddings.shape}, doc_embeddings: {doc_embeddings.shape}.")

    def score(self, query_embeddings, doc_embeddings, **kwargs):
        logits = self.compute_logits(query_embeddings, doc_embeddings)
        if self.temperature == "sqrt":
            return logits / np.sqrt(query_embeddings.shape[-1])
        return logits / self.temperature

    def get_encoder_output(self, args, **kwargs):
        assert len(args["input_ids"].shape) == 2
        outputs = self.encoder(**args, **kwargs)
        if isinstance(outputs, (tuple, list)):
            return outputs[0]
        return outputs

    def encoder_embed(self, args, **kwargs):
        shape = args["input_ids"].shape
        if len(shape) != 2:   # [..., S] -> [prod(...), S] and back
            args = {k: (v.reshape(-1, shape[-1]) if torch.is_tensor(v) else v) for k, v in args.items()}
        embeddings = self.get_encoder_output(args, **kwargs)
        if len(shape) != 2:
            embeddings = embeddings.reshape(*shape[:-1], embeddings.shape[-1])
        return embeddings

    def query(self, args, **kwargs):
        emb = self.encoder_embed(args, **kwargs)
        return torch.nn.functional.normalize(emb, dim=-1) if self.norm_query else emb

    def doc(self, args, **kwargs):
        emb = self.encoder_embed(args, **kwargs)
        return torch.nn.functional.normalize(emb, dim=-1) if self.norm_doc else emb

    def doc_packed(self, token_ids, seq_lens, max_len, total_tokens=None):
        """``doc`` from the ragged token list of a right-padded batch (``encoders.forward_packed``): the corpus-encode loop's feed
        (``kirag_amd.compute_corpus_embeddings``) ships tokens in this form; same rows, bit for bit, as ``doc({"input_ids", "attention_mask"})``."""
        emb = self.encoder.forward_packed(token_ids, seq_lens, max_len, total_tokens)
        return torch.nn.functional.normalize(emb, dim=-1) if self.norm_doc else emb

    def save_model(self, save_path):
        self.encoder.save_pretrained(save_path)

    def load_model(self, save_path):
        self.encoder = load_retriever(self.retriever_name, save_path, **self.retriever_kwargs, **self.kwargs)


class InBatchRetriever(BaseRetriever):

    def forward(self, query_args, doc_args, labels=None, **kwargs):
        q = self.query(query_args, **kwargs)
        gq = get_global_embeddings_for_inbatchtraining(self.local_rank, self.world_size, q)
        d = self.doc(doc_args, **kwargs)
        gd = get_global_embeddings_for_inbatchtraining(self.local_rank, self.world_size, d)
        glabels = get_global_labels_for_inbatchtraining(self.local_rank, self.world_size, labels, len(d))
        scores = self.score(gq, gd)
        if glabels is not None:
            loss = nn.CrossEntropyLoss()(scores, glabels)
            return (loss, scores, gq, gd)
        return (scores, gq, gd)


class DenseRetriever(nn.Module):

    def __init__(self, retriever: BaseRetriever, collator, indexer: Optional[Indexer] = None, corpus=None, batch_size: int = 4,
                 encode_batch_size: int = 256, **kwargs):
        super().__init__()
        self.retriever = retriever
        self.device = self.retriever.device
        self.retriever.eval()
        self.collator = collator
        self.indexer = indexer
        self.corpus = corpus
        self.batch_size = batch_size                 # kept for signature parity (reference default 4, retrieve.py passes 8)
        self.encode_batch_size = max(int(encode_batch_size), int(batch_size))
        self.kwargs = kwargs

    def get_documents(self, docid_list: Union[List[str], Dict[str, float]]) -> List[dict]:
        if isinstance(docid_list, list):
            return [deepcopy(self.corpus.get_document(docid)) for docid in docid_list]
        if isinstance(docid_list, dict):
            documents = []
            for docid, score in sorted(docid_list.items(), key=lambda kv: kv[1], reverse=True):
                document = deepcopy(self.corpus.get_document(docid))
                document["score"] = float(score)
                documents.append(document)
            return documents
        raise ValueError(f"{type(docid_list)} is not a supported type for \"docid_list\"!")

    def _embed(self, texts: List[str], which: str, max_length, verbose: bool, on_device: bool = False, **kwargs) -> Tensor:
        assert isinstance(texts, list) and len(texts) > 0   # must provide queries / documents
        encode = self.collator.encode_query if which == "query" else self.collator.encode_doc
        embed = self.retriever.query if which == "query" else self.retriever.doc
        chunks = []
        enc = getattr(self.retriever, "encoder", None)
        host_ok = bool(getattr(enc, "accepts_host_inputs", False)) and not getattr(enc, "training", True)
        for s in range(0, len(texts), self.encode_batch_size):
            inputs = encode(texts[s:s + self.encode_batch_size], max_length=max_length, **kwargs)
            if not host_ok:                                 # (retrievers.py:205 `to_device`; the HIP encoders upload the collator's CPU tensors themselves,
                inputs = to_device(inputs, self.device)     #  from pinned staging, asynchronously)
            chunks.append(embed(inputs).detach())
        out = chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=0)
        if on_device and out.is_cuda:
            return out                                      # batch_retrieve: the search reads the embeddings where they are; _check_inputs() follows it
        out = out.cpu()
        self._check_inputs()
        return out

    def _check_inputs(self) -> None:
        hip = getattr(getattr(self.retriever, "encoder", None), "_hip", None)
        if hip is not None:
            hip.check()                                     # deferred input errors (token ids outside the vocabulary, token_type_ids != 0) surface here

    def calculate_query_embeddings(self, queries: List[str], max_length: int = None, verbose: bool = False, **kwargs) -> Tensor:
        return self._embed(queries, "query", max_length, verbose, **kwargs)

    def calculate_document_embeddings(self, documents: List[str], max_length: int = None, verbose: bool = False, **kwargs) -> Tensor:
        return self._embed(documents, "doc", max_length, verbose, **kwargs)

    def parse_indexer_output(self, indexer_output):
        results = []
        for topk_str_indices, topk_score_array in indexer_output:
            one = []
            for docid, score in zip(topk_str_indices, topk_score_array):
                if self.corpus is not None:
                    document = deepcopy(self.corpus.get_document(docid))
                    document["score"] = float(score)
                else:
                    document = {"id": docid, "score": score}
                one.append(document)
            results.append(one)
        return results

    def batch_retrieve(self, queries: List[str], topk: int, verbose: bool = False, **kwargs) -> List[dict]:
        # (retrievers.py:198-199 goes embeddings -> .cpu().numpy() -> faiss.)  With the index resident on the encoder's own GPU the embeddings stay where the
        # encoder wrote them: one device round trip per hop less; same fp32 values, same results.
        if getattr(self.indexer, "accepts_device_queries", False):
            max_length = kwargs.pop("max_length", None)
            q = self._embed(queries, "query", max_length, verbose, on_device=True, **kwargs)
            if not q.is_cuda:
                q = q.numpy()
            knn = self.indexer.search_knn(query_vectors=q, top_docs=topk, index_batch_size=1024, verbose=verbose)
            self._check_inputs()
            return self.parse_indexer_output(knn)
        q = self.calculate_query_embeddings(queries=queries, verbose=verbose, **kwargs).numpy()
        knn = self.indexer.search_knn(query_vectors=q, top_docs=topk, index_batch_size=1024, verbose=verbose)
        return self.parse_indexer_output(knn)

    def forward(self, queries: Union[str, List[str]], topk: int, verbose: bool = False, **kwargs):
        assert self.indexer is not None   # must provide indexer
        if isinstance(queries, str):
            return self.batch_retrieve([queries], topk=topk, verbose=verbose, **kwargs)[0]
        return self.batch_retrieve(queries, topk=topk, verbose=verbose, **kwargs)
