"""Tokenizer worker process of ``compute_corpus_embeddings.pool_map`` (multi-process tokenisation).

Protocol on stdin / stdout, every message = u64 little-endian length + pickle:
  first message   the collator (``kirag_amd.collators.E5Collator`` / ``BGECollator`` with its HF tokenizer)
  then, per batch a list of passage strings -> reply ``(input_ids int32 [n, S], attention_mask uint8 [n, S])`` = ``collator.encode_doc(texts)``
                  (``dataset/collators.py:59-81,143-145`` semantics: prefix, pad to the longest of the batch, truncate at doc_maxlength),
                  or a string with the error message.
The process never touches the GPU and exits when stdin closes."""
import pickle
import struct
import sys


def main() -> None:
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                      # anything a library prints must not corrupt the protocol stream

    def recv():
        head = inp.read(8)
        if len(head) != 8:
            return None
        (n,) = struct.unpack("<Q", head)
        return pickle.loads(inp.read(n))

    def send(obj):
        b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        out.write(struct.pack("<Q", len(b))); out.write(b); out.flush()
    collator = recv()
    if collator is None:
        return
    import numpy as np
    while True:
        texts = recv()
        if texts is None:
            return
        try:
            enc = collator.encode_doc(texts)
            send((enc["input_ids"].numpy().astype(np.int32), enc["attention_mask"].numpy().astype(np.uint8)))
        except Exception as e:   # noqa: BLE001 - reported to the parent
            send(f"{type(e).__name__}: {e}")


if __name__ == "__main__":
    main()
