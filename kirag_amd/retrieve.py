"""MI355X counterpart of the retriever wiring of the reference's ``retrieve.py`` (``setup_retriever_model`` :86-118): same arguments
(``args.retriever_name / tokenizer_name_or_path / query_maxlength / doc_maxlength / retriever_model_name_or_path / local_rank / corpus /
index_folder / embedding_size / per_gpu_batch_size``), same return value ``(dense_retriever, corpus_dataset)``, with the encoder and the
flat inner-product index resident on the GPU.  The KiRAG model itself (``setup_kirag_model`` :120-143: LLM, KG generator) is out of scope and
stays in the reference; it only needs the ``dense_retriever`` returned here.

Extras over the reference (all optional, none changes results):
  * ``corpus_dataset=`` lets a caller pass its own corpus object instead of ``utils.const.CORPUS_MAP`` (which lives in the reference);
  * the ``retriever/e5.py`` singleton is pointed at the already resident e5 encoder when the retriever is an E5 one, so the
    exemplar ranking of ``knowledge_graph/models.py:1309-1317`` does not load a third copy of e5-large;
  * ``device=`` / ``args.local_rank`` select the GPU that holds the index shard and the encoder.
"""
from __future__ import annotations

import logging

from .collators import COLLATOR_MAP
from .retriever import e5 as e5_module
from .retriever.index import Indexer
from .retriever.retrievers import DenseRetriever, InBatchRetriever

logger = logging.getLogger(__file__)


def setup_retriever_model(args, corpus_dataset=None, tokenizer=None):
    """``retrieve.py:86-118`` on the HIP path."""
    retriever_name = args.retriever_name
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name_or_path)
    if tokenizer.pad_token is None or tokenizer.pad_token_id is None:
        logger.warning("Missing padding token, adding a new pad token!")
        tokenizer.add_special_tokens({"pad_token": '[PAD]'})
    collator = COLLATOR_MAP[retriever_name](tokenizer=tokenizer, query_maxlength=args.query_maxlength, doc_maxlength=args.doc_maxlength)
    retriever = InBatchRetriever(retriever_name=retriever_name, model_name_or_path=args.retriever_model_name_or_path,
                                 local_rank=args.local_rank, temperature=0.01)
    device = max(int(getattr(args, "local_rank", -1)), 0)
    retriever = retriever.to(f"cuda:{device}")
    retriever.eval()
    if corpus_dataset is None:
        logger.info(f"Loading corpus from {args.corpus} ...")
        try:   # corpus datasets are the reference's own (dataset/corpus.py, out of scope): needs the reference on PYTHONPATH
            from utils.const import CORPUS_MAP
        except ImportError as e:
            raise ImportError("args.corpus needs the KiRAG repository on PYTHONPATH (utils/const.py CORPUS_MAP), or pass corpus_dataset=") from e
        corpus_dataset = CORPUS_MAP[args.corpus](title_prefix="title: ", passage_prefix="text: ")
    logger.info(f"Loading index from {args.index_folder} ...")
    indexer = Indexer(args.embedding_size, metric="inner_product", device=device)
    indexer.deserialize_from(args.index_folder)
    dense_retriever = DenseRetriever(retriever=retriever, collator=collator, indexer=indexer, corpus=corpus_dataset,
                                     batch_size=args.per_gpu_batch_size)
    if retriever_name == "E5Retriever":
        e5_module.set_model(retriever.encoder, tokenizer)
    return dense_retriever, corpus_dataset
