"""kirag_amd — MI355X-native dense-retrieval core behind KiRAG's retriever surface.

Mirrors the reference's module layout for the hot path only:
  kirag_amd.retriever.index       <- retriever/index.py       (Indexer)
  kirag_amd.retriever.encoders    <- retriever/encoders.py    (E5Encoder, BGEEncoder)
  kirag_amd.retriever.retrievers  <- retriever/retrievers.py  (RETRIEVER_MAP, BaseRetriever, DenseRetriever ...)
  kirag_amd.retriever.e5          <- retriever/e5.py
All arithmetic runs in libkirag_amd.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
__version__ = "0.1.0"
