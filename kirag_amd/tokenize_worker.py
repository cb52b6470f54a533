"""Tokenizer worker process of ``kirag_amd.feed.TokenFeed`` (multi-process tokenisation for ``cal_doc_embeddings``).

Protocol on stdin / stdout (``kirag_amd.feed`` holds the other end and the frame layout):
  first message   u64 length + pickle of ``{"collator": E5Collator / BGECollator with its HF tokenizer, "vocab": int or None}``  ->  a READY frame
  then, per batch u64 length + pickle of the list of passage strings  ->  one binary FRAME (``feed.pack_frame``): the batch as ``collator.encode_doc(texts)``
                  yields it (``dataset/collators.py:59-81,143-145`` semantics: prefix, pad to the longest of the batch, truncate at doc_maxlength), but RAGGED -
                  int32 lengths + the int32 ids of the attended positions back to back, 4-16x fewer bytes than padded int64 ``input_ids`` + ``attention_mask``,
                  no pickle - or, for a tokenizer that does not pad on the right, the padded ids and mask; or an error frame.
Token ids are validated against ``vocab`` HERE (attended positions only), so the consumer thread of the encode loop does no per-batch reductions.
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

    from kirag_amd import feed
    try:
        init = recv()                            # un-pickling imports the collator's tokenizer class (transformers, torch): the slow part of the start
        if init is None:
            return
        collator, vocab = init["collator"], init.get("vocab")
    except Exception as e:   # noqa: BLE001 - reported to the parent, which goes on without this worker
        for part in feed.error_frame(f"cannot load the collator: {type(e).__name__}: {e}"):
            out.write(part)
        out.flush()
        return
    for part in feed.ready_frame():
        out.write(part)
    out.flush()
    while True:
        texts = recv()
        if texts is None:
            return
        try:
            for part in feed.pack_frame(feed.tokenize_batch(collator, texts), vocab):
                out.write(part)
        except Exception as e:   # noqa: BLE001 - reported to the parent
            for part in feed.error_frame(f"{type(e).__name__}: {e}"):
                out.write(part)
        out.flush()


if __name__ == "__main__":
    main()
