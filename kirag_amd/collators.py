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
        rows = self._fast_rows(text_list, maxlength, padding) if not kwargs or set(kwargs) <= {"max_length"} else None
        if rows is not None:
            return self._pad_rows(rows, maxlength, padding)
        return self._encode_reference(text_list, maxlength, padding)

    def _encode_reference(self, text_list, maxlength, padding):
        """The reference's call, verbatim (dataset/collators.py:76-80)."""
        pad = "max_length" if padding == "max_length" else True      # True = pad to the longest sequence of the batch
        enc = self.tokenizer(text_list, max_length=maxlength, padding=pad, truncation=True, return_tensors="pt")
        return {"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"]}

    # ---- fast path: the SAME token ids without transformers' per-call Python (BatchEncoding, list flattening, tensor conversion) ------------------------------
    # `tokenizer(texts, max_length, padding, truncation=True, return_tensors="pt")` of a fast tokenizer is: set truncation / padding on the Rust tokenizer, its
    # `encode_batch`, then ~0.2 ms (one query) to ~5 ms (512 passages) of pure-Python conversion.  The fast path calls the Rust tokenizer itself and pads with
    # numpy.  It is taken only for right-padding fast tokenizers and only after its output has been compared, once per (max_length, padding), with the
    # reference call on the very batch it is first used for — any difference switches it off for good (this tokenizer / transformers version then simply keeps
    # the reference call).  One KiRAG hop: tokenizer + collator 0.39 -> 0.2 ms (bench.py latency.kirag_hop_nq1_surface).
    def _fast_rows(self, text_list, maxlength, padding):
        if getattr(self, "_fast_off", False):
            return None
        tok = self.tokenizer
        bt = getattr(tok, "_tokenizer", None)
        if bt is None or not getattr(tok, "is_fast", False) or getattr(tok, "padding_side", "right") != "right" or getattr(tok, "pad_token_id", None) is None \
                or not hasattr(bt, "encode_batch") or maxlength is None:
            return None
        if isinstance(text_list, str):
            text_list = [text_list]
        try:
            tr = bt.truncation
            if tr is None or tr.get("max_length") != maxlength or tr.get("stride", 0) != 0 or tr.get("strategy") != "longest_first" \
                    or tr.get("direction", "right") != getattr(tok, "truncation_side", "right"):
                bt.enable_truncation(max_length=int(maxlength), stride=0, strategy="longest_first", direction=getattr(tok, "truncation_side", "right"))
            if bt.padding is not None:
                bt.no_padding()
            rows = [e.ids for e in bt.encode_batch(list(text_list), add_special_tokens=True)]
        except Exception:   # noqa: BLE001 - an unexpected backend: keep the reference call
            self._fast_off = True
            return None
        key = (int(maxlength), padding)
        checked = self.__dict__.setdefault("_fast_checked", set())
        if key not in checked:
            ref = self._encode_reference(text_list, maxlength, padding)
            got = self._pad_rows(rows, maxlength, padding)
            import torch
            if not (torch.equal(ref["input_ids"], got["input_ids"]) and torch.equal(ref["attention_mask"], got["attention_mask"])
                    and ref["input_ids"].dtype == got["input_ids"].dtype):
                import logging
                logging.getLogger(__name__).warning("the collator's fast tokenizer path does not reproduce tokenizer(...) for this tokenizer: switched off")
                self._fast_off = True
                return None
            checked.add(key)
        return rows

    def _pad_rows(self, rows, maxlength, padding):
        import numpy as np
        import torch
        n = len(rows)
        L = int(maxlength) if padding == "max_length" else max(len(r) for r in rows)
        ids = np.full((n, L), int(self.tokenizer.pad_token_id), dtype=np.int64)
        mask = np.zeros((n, L), dtype=np.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
            mask[i, :len(r)] = 1
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}

    def encode_doc_ragged(self, doc_list, **kwargs):
        """``encode_doc`` without the padding: ``(token rows, padded width)`` = the attended ids of every passage as lists + the width the padded batch would
        have — what ``kirag_amd.feed`` ships to ``kr_encoder_forward_packed``.  None when the fast path is not available (the caller pads and strips instead)."""
        maxlen = kwargs.get("max_length", None) or self.doc_maxlength
        if isinstance(doc_list, (tuple, list)) and len(doc_list) == 0:
            raise ValueError("text_list is None or an empty tuple/list!")
        rows = self._fast_rows([self.doc_prefix + d for d in doc_list], maxlen, self.doc_padding)
        if rows is None:
            return None
        return rows, (int(maxlen) if self.doc_padding == "max_length" else max(len(r) for r in rows))

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
