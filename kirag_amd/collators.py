"""Tokenisation + prefixing contract of the reference's ``dataset/collators.py`` for the two retrievers on the path
(``E5Collator`` :132-145, ``BGECollator`` :148-156, ``RetrieverCollator.encode`` :59-89).  The HF tokenizer itself stays a
host-side dependency (out of scope to rewrite); this module only reproduces the strings and padding
policy that define the encoder's input, so that the entry points below work without the reference on PYTHONPATH.
Pinned by tests/golden/g4_g8_retriever.npz (G7)."""
from __future__ import annotations

BGE_QUERY_INSTRUCTION = "Represent this sentence for searching relevant passages:"


class RetrieverCollator:
    query_prefix = ""
    doc_prefix = ""

    def __init__(self, tokenizer, query_maxlength, doc_maxlength=None, query_padding="max_sequence", doc_padding="max_sequence", **kwargs):
        self.tokenizer = tokenizer
        self.query_maxlength = query_maxlength
        self.doc_maxlength = query_maxlength if doc_maxlength is None else doc_maxlength
        self.query_padding, self.doc_padding = query_padding, doc_padding
        self.kwargs = kwargs

    def encode(self, text_list, maxlength, padding, **kwargs):
        if padding not in ("max_length", "max_sequence"):
            raise AssertionError("padding must be chosen from [\"max_length\", \"max_sequence\"]")
        if text_list is None or (isinstance(text_list, (tuple, list)) and len(text_list) == 0):
            raise ValueError("text_list is None or an empty tuple/list!")
        if not (isinstance(text_list, str) or isinstance(text_list[0], str)):
            raise ValueError("only flat lists of strings are supported on the MI355X path "
                             "(nested question/passage lists are used by the reference's training collator only)")
        pad = "max_length" if padding == "max_length" else True      # True = pad to the longest sequence of the batch
        enc = self.tokenizer(text_list, max_length=maxlength, padding=pad, truncation=True, return_tensors="pt")
        return {"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"]}

    def encode_query(self, query_list, **kwargs):
        maxlen = kwargs.get("max_length", None) or self.query_maxlength
        return self.encode([self.query_prefix + q for q in query_list], maxlen, self.query_padding, **kwargs)

    def encode_doc(self, doc_list, **kwargs):
        maxlen = kwargs.get("max_length", None) or self.doc_maxlength
        return self.encode([self.doc_prefix + d for d in doc_list], maxlen, self.doc_padding, **kwargs)


class E5Collator(RetrieverCollator):
    query_prefix = "query: "       # collators.py:139-141
    doc_prefix = "passage: "       # collators.py:143-145


class BGECollator(RetrieverCollator):
    query_prefix = BGE_QUERY_INSTRUCTION + " "   # collators.py:153-156; documents are not prefixed


COLLATOR_MAP = {"E5Retriever": E5Collator, "BGERetriever": BGECollator}   # utils/const.py:5-8
