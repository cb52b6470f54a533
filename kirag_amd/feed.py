"""Tokenizer feed of the corpus-encode loop (replaces the reference's single-threaded DataLoader, ``utils/utils.py:118-126`` ``num_workers=0``,
in front of ``compute_corpus_embeddings.py:77-81``): batches of passage TEXT become token frames sitting in pinned host memory, in corpus order, ahead of the GPU.

What travels is RAGGED: the int32 ids of the attended positions of every sequence back to back + one int32 length per sequence - exactly the token list
``kr_encoder_forward_packed`` packs on the device (``k_rag_fill``), 4-16x fewer bytes than the padded int64 ``input_ids`` + ``attention_mask`` the collator
returns, as raw buffers (a 32-byte header + two arrays), never a pickle.  A batch whose mask is not "1^len 0^(S-len)" (a tokenizer that pads on the left)
travels padded instead and takes the ``model.doc({"input_ids", "attention_mask"})`` route; results are the same rows either way.

``TokenFeed`` tokenises on one background thread of this process (HF fast tokenizers release the GIL and bring their own thread pool) and, when
``workers > 0``, in that many PROCESSES as well (``python -m kirag_amd.tokenize_worker``).  Batches are handed out dynamically - whoever is free takes the
next one - so the local thread carries the stream while the workers are still starting (importing torch + transformers takes seconds: a static split made a
100 k-passage run 25 % slower than no workers at all) and the workers add capacity on hosts where one tokenizer cannot keep the encoder fed.  Every worker
has a serve thread here that builds the batch's strings (dataset access stays in the parent), sends them, and reads the reply STRAIGHT INTO a slot of a
persistent pinned ring (``readinto``: no intermediate bytes object, no per-batch ``pin_memory()`` allocation, the GIL is released while the pipe is read).
Token ids are validated where they are produced (``pack_frame``), so the consumer - the one Python thread that launches the encoder - does nothing per batch
but take the next frame.  Slot j % R serves batch j and is handed back by ``release`` once the upload that read it has completed; batch j may be started
only when j < released + R, and batches are started in increasing order, so the ring cannot deadlock (the oldest outstanding batch always has its slot).
"""
from __future__ import annotations

import os
import pickle
import struct
import subprocess
import sys
import threading
from typing import Callable, Iterable, List, Optional, Tuple

import itertools
import logging

import numpy as np

logger = logging.getLogger(__name__)
FRAME_MAGIC = 0x4654524B          # "KRTF"
KIND_RAGGED, KIND_PADDED, KIND_ERROR, KIND_BAD_ID, KIND_READY = 0, 1, 2, 3, 4
_HEAD = struct.Struct("<IIIIqQ")  # magic, kind, n sequences, S (padded width), T (tokens | n*S | message bytes), aux (KIND_BAD_ID: (min << 32) | max as two int32)


class Tokens:
    """One tokenised batch, ragged or padded (numpy, host).  ``ids``/``lens`` int32; ``mask`` uint8 [n*S] only when padded."""
    __slots__ = ("kind", "n", "S", "T", "ids", "lens", "mask")

    def __init__(self, kind, n, S, T, ids, lens=None, mask=None):
        self.kind, self.n, self.S, self.T, self.ids, self.lens, self.mask = kind, n, S, T, ids, lens, mask


def tokens_of(enc) -> Tokens:
    """The collator's ``{"input_ids", "attention_mask"}`` ([n,S] integer tensors / arrays) as ``Tokens``: ragged when every mask row is 1^len 0^(S-len)."""
    ids, mask = enc["input_ids"], enc["attention_mask"]
    ids = ids.numpy() if hasattr(ids, "numpy") else np.asarray(ids)
    mask = mask.numpy() if hasattr(mask, "numpy") else np.asarray(mask)
    if ids.ndim != 2 or ids.shape != mask.shape:
        raise ValueError(f"input_ids {ids.shape} / attention_mask {mask.shape}: expected two [n, S] arrays")
    n, S = ids.shape
    on = mask != 0
    lens = on.sum(axis=1, dtype=np.int32)
    if ids.dtype.itemsize > 4:     # narrowing to int32 must not wrap an out-of-range id into the vocabulary: saturate, the range check then rejects it
        ids = np.clip(ids, -2**31, 2**31 - 1)
    if np.array_equal(on, np.arange(S, dtype=np.int32)[None, :] < lens[:, None]):
        rag = ids[on].astype(np.int32, copy=False)
        return Tokens(KIND_RAGGED, n, S, int(rag.size), np.ascontiguousarray(rag), np.ascontiguousarray(lens))
    return Tokens(KIND_PADDED, n, S, n * S, np.ascontiguousarray(ids, dtype=np.int32).reshape(-1), np.ascontiguousarray(lens),
                  np.ascontiguousarray(on, dtype=np.uint8).reshape(-1))


def tokens_from_rows(rows, S: int) -> Tokens:
    """``Tokens`` (ragged) from per-sequence id lists (``collator.encode_doc_ragged``): no padded [n,S] arrays are ever built."""
    lens = np.fromiter((len(r) for r in rows), dtype=np.int32, count=len(rows))
    flat = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int64, count=int(lens.sum()))
    flat = np.clip(flat, -2**31, 2**31 - 1).astype(np.int32)
    return Tokens(KIND_RAGGED, len(rows), int(S), int(flat.size), flat, lens)


def tokenize_batch(collator, texts) -> Tokens:
    """One batch of passage strings -> ``Tokens``: through the collator's ragged fast path when it has one, else ``encode_doc`` (padded) stripped of its padding."""
    ragged = getattr(collator, "encode_doc_ragged", None)
    if ragged is not None:
        out = ragged(texts)
        if out is not None:
            return tokens_from_rows(*out)
    return tokens_of(collator.encode_doc(texts))


def attended_range(t: Tokens) -> Optional[Tuple[int, int]]:
    """(min, max) token id over the ATTENDED positions (the only ones the HIP forward reads), None for a batch without any."""
    a = t.ids if t.kind == KIND_RAGGED else t.ids[t.mask != 0]
    return (int(a.min()), int(a.max())) if a.size else None


def pack_frame(t: Tokens, vocab: Optional[int] = None) -> List[bytes]:
    """The byte strings of one frame (header, then the arrays as they lie in memory).  With ``vocab``, an id outside [0, vocab) at an attended position
    turns the frame into a KIND_BAD_ID frame carrying (min, max) instead of the tokens: nothing of such a batch is encoded, indexed or written."""
    if vocab is not None:
        r = attended_range(t)
        if r is not None and (r[0] < 0 or r[1] >= vocab):
            lo, hi = r
            return [_HEAD.pack(FRAME_MAGIC, KIND_BAD_ID, t.n, t.S, 0, ((lo & 0xffffffff) << 32) | (hi & 0xffffffff))]
    parts = [_HEAD.pack(FRAME_MAGIC, t.kind, t.n, t.S, t.T, 0), t.lens.tobytes(), t.ids.tobytes()]
    if t.kind == KIND_PADDED:
        parts.append(t.mask.tobytes())
    return parts


def error_frame(message: str) -> List[bytes]:
    b = message.encode("utf-8", "replace")
    return [_HEAD.pack(FRAME_MAGIC, KIND_ERROR, 0, 0, len(b), 0), b]


def ready_frame() -> List[bytes]:
    """Sent once by a worker when its collator is loaded: its serve thread takes batches only from then on."""
    return [_HEAD.pack(FRAME_MAGIC, KIND_READY, 0, 0, 0, 0)]


def bad_id_message(first_passage_id, vocab, lo, hi) -> str:
    return (f"input_ids of the batch starting at passage id {first_passage_id} contain a token id outside [0, {vocab}) "
            f"(min {lo}, max {hi}): nothing of this batch was encoded, indexed or written")


class Slot:
    """One entry of the pinned ring: room for a batch's lengths and tokens (and, for padded frames, its mask)."""

    def __init__(self, max_seqs: int, max_tokens: int, pinned: bool):
        import torch
        self.pinned = pinned
        self.ids_t = torch.empty(max(1, max_tokens), dtype=torch.int32, pin_memory=pinned)
        self.lens_t = torch.empty(max(1, max_seqs), dtype=torch.int32, pin_memory=pinned)
        self.mask_t = None
        self.ids, self.lens = self.ids_t.numpy(), self.lens_t.numpy()

    def fit(self, n: int, T: int, padded: bool):
        import torch
        if T > self.ids.size:      # a batch larger than announced (a tokenizer ignoring max_length): grow this slot once, pinned like the rest
            self.ids_t = torch.empty(T, dtype=torch.int32, pin_memory=self.pinned); self.ids = self.ids_t.numpy()
        if n > self.lens.size:
            self.lens_t = torch.empty(n, dtype=torch.int32, pin_memory=self.pinned); self.lens = self.lens_t.numpy()
        if padded and (self.mask_t is None or self.mask_t.numel() < T):
            self.mask_t = torch.empty(T, dtype=torch.uint8, pin_memory=self.pinned)


class Frame:
    """What ``TokenFeed`` yields: batch ``index`` (position in ``items``), its ``doc_ids``, and the tokens as int32 TENSORS over a pinned ring slot —
    ``kind == KIND_RAGGED``: ``ids[:T]``, ``lens[:n]`` for ``model.doc_packed(ids, lens, S, T)``; ``KIND_PADDED``: ``inputs()`` builds the int64 [n,S] pair.
    Call ``feed.release(frame, event)`` when the work that reads the slot has been enqueued (``event.synchronize()`` is awaited before the slot is reused)."""
    __slots__ = ("index", "doc_ids", "kind", "n", "S", "T", "slot")

    @property
    def ids(self):
        return self.slot.ids_t

    @property
    def lens(self):
        return self.slot.lens_t[:self.n]

    def inputs(self, pad_id: int = 0):
        """The padded int64 ``{"input_ids", "attention_mask"}`` [n,S] of this frame (host tensors; a copy; ``pad_id`` fills the masked positions of a ragged frame)."""
        import torch
        if self.kind == KIND_PADDED:
            ids = self.slot.ids_t[:self.T].view(self.n, self.S).to(torch.int64)
            return {"input_ids": ids, "attention_mask": self.slot.mask_t[:self.T].view(self.n, self.S).to(torch.int64)}
        return repad(self.slot.ids[:self.T], self.slot.lens[:self.n], self.S, pad_id)


def repad(ids: np.ndarray, lens: np.ndarray, S: int, pad_id: int):
    """Ragged tokens back to the collator's right-padded int64 ``{"input_ids", "attention_mask"}`` [n,S] (tests, the ``pool_map`` compatibility wrapper)."""
    import torch
    n = len(lens)
    on = np.arange(S, dtype=np.int32)[None, :] < np.asarray(lens, dtype=np.int32)[:, None]
    out = np.full((n, S), pad_id, dtype=np.int64)
    out[on] = ids
    return {"input_ids": torch.from_numpy(out), "attention_mask": torch.from_numpy(on.astype(np.int64))}


def _read_exact(f, view: memoryview) -> None:
    got = 0
    while got < len(view):
        k = f.readinto(view[got:])
        if not k:
            raise RuntimeError("tokenizer worker exited unexpectedly")
        got += k


class TokenFeed:
    """Ordered stream of token ``Frame``s for ``items`` (see the module docstring).

    ``make_texts(item) -> (texts, doc_ids)`` builds a batch's strings in this process; ``collator.encode_doc`` tokenises them (on the local thread or in a
    worker process).  ``max_seqs`` / ``max_len``: the largest batch / padded width to expect (slots are sized for it and grow if a batch exceeds it).
    ``vocab``: validate attended ids against [0, vocab); a violation raises ``ValueError`` at the consumer when that batch's turn comes.
    ``depth``: frames buffered beyond one per tokenizer.  ``local=False``: worker processes only (tests).  Iterate once; ``close()`` (or exhausting /
    abandoning the iterator) stops threads and processes."""

    def __init__(self, make_texts: Callable, collator, items: Iterable, workers: int, depth: int, max_seqs: int, max_len: int, vocab: Optional[int] = None,
                 pinned: bool = False, local: bool = True):
        self.make_texts, self.collator, self.items = make_texts, collator, list(items)
        self.vocab = int(vocab) if vocab else None
        self.workers = max(0, min(int(workers), len(self.items)))
        self.local = bool(local) or self.workers == 0
        self.R = max(3, self.workers + int(self.local) + max(1, int(depth)) + 2)
        self.slots = [Slot(max_seqs, max_seqs * max_len, pinned) for _ in range(min(self.R, max(1, len(self.items))))]
        self.R = len(self.slots) if len(self.slots) < self.R else self.R
        self.cond = threading.Condition()
        self.ready = {}             # batch index -> Frame | BaseException
        self.next_j = 0             # next batch to hand to a tokenizer
        self.released = 0           # every batch < released has given its slot back
        self.pending_release = []   # (batch index, event) in order
        self.stop = False
        self.made_by = {"local": 0, "workers": 0}   # batches per producer kind (diagnostics)
        self.procs: List[subprocess.Popen] = []
        self.threads: List[threading.Thread] = []
        self._started = False

    # ---- producer side -------------------------------------------------------------------------------------------------------------------------------
    def _take(self) -> int:
        """The next batch index for a free tokenizer (-1: none left / stopping); returns once its ring slot is free."""
        with self.cond:
            if self.stop or self.next_j >= len(self.items):
                return -1
            j = self.next_j
            self.next_j += 1
            self.cond.wait_for(lambda: self.stop or j < self.released + self.R)
            return -1 if self.stop else j

    def _post(self, j: int, what) -> None:
        with self.cond:
            self.ready[j] = what
            self.cond.notify_all()

    def _frame(self, j: int, doc_ids, kind, n, S, T) -> Frame:
        f = Frame()
        f.index, f.doc_ids, f.kind, f.n, f.S, f.T, f.slot = j, doc_ids, kind, n, S, T, self.slots[j % self.R]
        return f

    def _serve_local(self) -> None:
        """Tokenise on this background thread (HF fast tokenizers release the GIL)."""
        j = -1
        try:
            while True:
                j = self._take()
                if j < 0:
                    return
                texts, doc_ids = self.make_texts(self.items[j])
                t = tokenize_batch(self.collator, texts)
                if self.vocab is not None:
                    r = attended_range(t)
                    if r is not None and (r[0] < 0 or r[1] >= self.vocab):
                        self._post(j, ValueError(bad_id_message(doc_ids[0] if doc_ids else "?", self.vocab, r[0], r[1])))
                        continue
                slot = self.slots[j % self.R]
                slot.fit(t.n, t.T, t.kind == KIND_PADDED)
                slot.lens[:t.n] = t.lens
                slot.ids[:t.T] = t.ids
                if t.kind == KIND_PADDED:
                    slot.mask_t.numpy()[:t.T] = t.mask
                self.made_by["local"] += 1
                self._post(j, self._frame(j, doc_ids, t.kind, t.n, t.S, t.T))
        except BaseException as e:   # noqa: BLE001 - forwarded to the consumer (as the failure of the batch this thread held)
            self._fail(j, e)

    def _fail(self, j: int, e: BaseException) -> None:
        with self.cond:
            if not self.stop:
                if j >= 0:
                    self.ready.setdefault(j, e)
                else:
                    self.ready.setdefault(-1, e)
            self.cond.notify_all()

    def _serve_worker(self, w: int, blob: bytes) -> None:
        p = self.procs[w]
        j = -1

        def send(payload: bytes) -> None:
            p.stdin.write(struct.pack("<Q", len(payload))); p.stdin.write(payload); p.stdin.flush()

        def header():
            _read_exact(p.stdout, memoryview(head))
            magic, kind, n, S, T, aux = _HEAD.unpack(head)
            if magic != FRAME_MAGIC:
                raise RuntimeError("tokenizer worker: corrupt frame header (did a library write to the worker's stdout?)")
            if kind == KIND_ERROR:
                msg = bytearray(T); _read_exact(p.stdout, memoryview(msg))
                raise RuntimeError("tokenizer worker: " + msg.decode("utf-8", "replace"))
            return kind, n, S, T, aux
        try:
            head = bytearray(_HEAD.size)
            send(blob)
            try:
                if header()[0] != KIND_READY:        # the worker has imported its libraries and un-pickled the collator: only now does it take batches
                    raise RuntimeError("tokenizer worker: expected the ready frame")
            except Exception as e:   # noqa: BLE001 - a worker that cannot start is a lost speed-up, not a lost result, as long as the local thread runs
                if not self.local:
                    raise
                logger.warning("tokenizer worker %d did not start (%s); continuing with the remaining tokenizers", w, e)
                return
            while True:
                j = self._take()
                if j < 0:
                    return
                texts, doc_ids = self.make_texts(self.items[j])
                send(pickle.dumps(texts, protocol=pickle.HIGHEST_PROTOCOL))
                kind, n, S, T, aux = header()
                if kind == KIND_BAD_ID:
                    lo, hi = struct.unpack("<ii", struct.pack("<II", (aux >> 32) & 0xffffffff, aux & 0xffffffff))
                    self._post(j, ValueError(bad_id_message(doc_ids[0] if doc_ids else "?", self.vocab, lo, hi)))
                    continue
                slot = self.slots[j % self.R]
                slot.fit(n, T, kind == KIND_PADDED)
                _read_exact(p.stdout, memoryview(slot.lens[:n]).cast("B"))
                _read_exact(p.stdout, memoryview(slot.ids[:T]).cast("B"))
                if kind == KIND_PADDED:
                    _read_exact(p.stdout, memoryview(slot.mask_t.numpy()[:T]).cast("B"))
                if n != len(doc_ids):
                    raise RuntimeError(f"tokenizer worker returned {n} sequences for a batch of {len(doc_ids)}")
                self.made_by["workers"] += 1
                self._post(j, self._frame(j, doc_ids, kind, n, S, T))
        except BaseException as e:   # noqa: BLE001 - forwarded to the consumer
            if isinstance(e, OSError):                          # a broken pipe: the worker is gone
                e = RuntimeError(f"tokenizer worker exited unexpectedly ({type(e).__name__}: {e})")
            self._fail(j, e)

    def _start(self) -> None:
        self._started = True
        self.threads = [threading.Thread(target=self._serve_local, daemon=True, name="kirag-amd-tokenize")] if self.local else []
        if self.workers > 0:
            env = dict(os.environ, TOKENIZERS_PARALLELISM="false", PYTHONPATH=os.pathsep.join(
                [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
            try:
                blob = pickle.dumps({"collator": self.collator, "vocab": self.vocab}, protocol=pickle.HIGHEST_PROTOCOL)
            except Exception as e:   # noqa: BLE001 - e.g. a tokenizer object that cannot be pickled
                if not self.local:
                    raise
                logger.warning("the collator cannot be sent to tokenizer processes (%s); tokenising in-process only", e)
                self.workers = 0
        if self.workers > 0:
            # plain child processes: no fork of a process that holds a GPU context and a tokenizer thread pool, no re-import of the caller's __main__
            self.procs = [subprocess.Popen([sys.executable, "-m", "kirag_amd.tokenize_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                          for _ in range(self.workers)]
            self.threads += [threading.Thread(target=self._serve_worker, args=(w, blob), daemon=True, name=f"kirag-amd-feed-{w}") for w in range(self.workers)]
        for t in self.threads:
            t.start()

    # ---- consumer side -------------------------------------------------------------------------------------------------------------------------------
    def release(self, frame: Frame, event=None) -> None:
        """The slot of ``frame`` may be reused once ``event`` (anything with ``synchronize()``; None = now) has completed.  Frames are released in order."""
        self.pending_release.append((frame.index, event))

    def _reap(self, keep: int) -> None:
        """Give back the slots of all but the ``keep`` most recently released frames (waits for their events: bounds how far the host runs ahead of the GPU)."""
        while len(self.pending_release) > keep:
            j, ev = self.pending_release.pop(0)
            if ev is not None:
                ev.synchronize()
            with self.cond:
                self.released = max(self.released, j + 1)
                self.cond.notify_all()

    def __iter__(self):
        if self._started:
            raise RuntimeError("a TokenFeed can be iterated once")
        self._start()
        try:
            for j in range(len(self.items)):
                self._reap(1)
                with self.cond:
                    self.cond.wait_for(lambda: j in self.ready or -1 in self.ready)
                    out = self.ready.pop(j, None)
                    if out is None:
                        out = self.ready[-1]
                if isinstance(out, BaseException):
                    raise out
                yield out
                if not self.pending_release or self.pending_release[-1][0] != j:
                    self.release(out)                      # the consumer did not say: assume it is done with the slot
            self._reap(0)
        finally:
            self.close()

    def close(self) -> None:
        with self.cond:
            self.stop = True
            self.cond.notify_all()
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        for p in self.procs:
            for f in (p.stdout,):
                try:
                    f.close()
                except Exception:
                    pass
        self.procs = []
