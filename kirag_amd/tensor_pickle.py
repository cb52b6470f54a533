"""``corpus_embeddings_{s}_{e}.pkl`` written as a STREAM (compute_corpus_embeddings.py:101-115 of the reference does ``pickle.dump(tensor, f)`` on a tensor it grew by
``torch.cat`` on the host).

The file the reference's readers open (``faiss_index_corpus.py:38``: ``pickle.load`` -> fp32 ``torch.Tensor [n, hidden]``) is, byte for byte, a small pickle
around one long run of raw row data:

    PROTO | GLOBAL torch._utils._rebuild_tensor_v2 | MARK
          | GLOBAL torch.storage._load_from_bytes | BINBYTES8 <L> <blob> | TUPLE1 REDUCE          -> the storage
          | 0 | (n, d) | (d, 1) | False | OrderedDict() | TUPLE REDUCE STOP
    blob  = torch's legacy storage container: pickle(magic) pickle(1001) pickle(sys_info) pickle(persistent id of the storage) pickle([key])
            int64 numel | numel * 4 bytes of rows

Everything except the rows is known as soon as the row count is (``n = min(rows per file, rows left)``), so the writer emits the head when the first batch
arrives, appends each batch's rows straight from the pinned landing buffer, and emits the 40-byte tail at the end: no [n, hidden] staging tensor on the host
(4 GiB per 1M-row file, twice that inside ``pickle.dump``), no multi-second ``pickle.dump`` holding the GIL while the encoder waits, and nothing left to do
after the last batch (the buffered form cost 8 % of a one-file run: profiles/r06/feed_probe.txt).

The layout is torch's, not ours, so it is VERIFIED against the running torch before it is used (``selftest``): a tiny stream must ``pickle.load`` to an equal
tensor with exactly-sized storage, and its storage blob head must equal, byte for byte, what ``torch.save(storage, _use_new_zipfile_serialization=False)``
itself writes for the same storage.  If either check fails (a future torch changes its tensor pickle) ``available()`` is False and ``_ShardWriter`` keeps the
buffered ``pickle.dump``.  Readers need nothing: the stream IS a tensor pickle (loads with torch >= 1.x, including the reference's pin, torch 2.2.1)."""
from __future__ import annotations

import collections
import io
import os
import pickle
import struct
from typing import Optional

import numpy as np

_MAGIC = 0x1950A86A20F9469CFC6C
_PROTOCOL_VERSION = 1001
_MARK = b"\x00kirag-amd-storage-bytes-marker\x00"


def _storage_blob_head(numel: int, key: str = "0") -> bytes:
    """torch.serialization._legacy_save of ONE fp32 cpu storage, up to and including the int64 element count in front of the raw data."""
    import torch
    sentinel = object()

    class _P(pickle.Pickler):
        def persistent_id(self, obj):   # noqa: D401 - torch's tuple form: ('storage', storage class, key, location, numel, view metadata)
            return ("storage", torch.FloatStorage, key, "cpu", numel, None) if obj is sentinel else None
    b = io.BytesIO()
    pickle.dump(_MAGIC, b, protocol=2)
    pickle.dump(_PROTOCOL_VERSION, b, protocol=2)
    pickle.dump(dict(protocol_version=_PROTOCOL_VERSION, little_endian=True, type_sizes=dict(short=2, int=4, long=4)), b, protocol=2)
    _P(b, protocol=2).dump(sentinel)
    pickle.dump([key], b, protocol=2)
    b.write(struct.pack("<q", numel))
    return b.getvalue()


def _envelope(rows: int, d: int):
    """(head, tail) of the outer pickle around the storage blob."""
    import torch

    class _S:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (_MARK,))

    class _T:
        def __reduce__(self):
            return (torch._utils._rebuild_tensor_v2, (_S(), 0, (rows, d), (d, 1), False, collections.OrderedDict()))
    raw = pickle.dumps(_T(), protocol=3)                                 # protocol 3: no FRAME opcodes to keep consistent
    tok = b"C" + bytes([len(_MARK)]) + _MARK                             # SHORT_BINBYTES <len> <marker>
    pos = raw.find(tok)
    if pos < 0 or raw.find(tok, pos + 1) >= 0 or raw[:2] != b"\x80\x03":
        raise RuntimeError("unexpected pickle layout")
    return b"\x80\x04" + raw[2:pos], raw[pos + len(tok):]               # PROTO 4: the storage bytes go in as BINBYTES8 (a protocol-4 opcode)


class StreamingTensorPickle:
    """A file that ``pickle.load``s to a contiguous float32 ``torch.Tensor [rows, d]``, written incrementally: ``append`` row blocks until ``rows`` rows are in,
    then ``close``.  The data goes to ``path + '.tmp'`` and is renamed on ``close``; ``abort`` removes it."""

    def __init__(self, path: str, rows: int, d: int):
        if rows <= 0 or d <= 0:
            raise ValueError("rows and d must be positive")
        self.path, self.rows, self.d, self.written = path, int(rows), int(d), 0
        head, self._tail = _envelope(self.rows, self.d)
        blob_head = _storage_blob_head(self.rows * self.d)
        self._f = open(path + ".tmp", "wb")
        self._f.write(head + b"\x8e" + struct.pack("<Q", len(blob_head) + self.rows * self.d * 4) + blob_head)

    def append(self, x: np.ndarray) -> None:
        if x.dtype != np.float32 or x.ndim != 2 or x.shape[1] != self.d or not x.flags.c_contiguous:
            raise ValueError(f"expected a C-contiguous float32 [m, {self.d}] array, got {x.dtype} {x.shape}")
        if self.written + x.shape[0] > self.rows:
            raise ValueError(f"{self.written} + {x.shape[0]} rows exceed the {self.rows} announced")
        self._f.write(memoryview(x).cast("B"))
        self.written += x.shape[0]

    def close(self) -> None:
        if self.written != self.rows:
            self.abort()
            raise RuntimeError(f"{self.written} of the {self.rows} announced rows were written")
        self._f.write(self._tail)
        self._f.close()
        os.replace(self.path + ".tmp", self.path)

    def abort(self) -> None:
        try:
            self._f.close()
        finally:
            try:
                os.remove(self.path + ".tmp")
            except OSError:
                pass


_OK: Optional[bool] = None


def selftest() -> bool:
    """The two checks of the module docstring against the running torch (a 3 x 4 tensor, in memory / a temp file)."""
    import tempfile
    import torch
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4) * 0.5 - 1.0
    # (1) torch's own storage container head for this storage == ours (the key is torch's choice: the storage's address)
    b = io.BytesIO()
    torch.save(t._typed_storage(), b, _use_new_zipfile_serialization=False)      # what TypedStorage.__reduce__ (the tensor pickle's storage) writes
    theirs = b.getvalue()
    key = str(t.untyped_storage()._cdata)
    ours = _storage_blob_head(12, key)
    if theirs[:len(ours)] != ours or len(theirs) != len(ours) + 48 or theirs[len(ours):] != t.numpy().tobytes():
        return False
    # (2) a stream loads as that tensor
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "t.pkl")
        w = StreamingTensorPickle(p, 3, 4)
        w.append(t.numpy()[:2]); w.append(t.numpy()[2:])
        w.close()
        with open(p, "rb") as f:
            back = pickle.load(f)
        whole = pickle.loads(pickle.dumps(t))
    return bool(isinstance(back, torch.Tensor) and back.dtype == torch.float32 and back.shape == t.shape and back.stride() == whole.stride()
                and back.is_contiguous() and back.untyped_storage().nbytes() == 48 and torch.equal(back, t) and not back.requires_grad)


def available() -> bool:
    global _OK
    if _OK is None:
        try:
            _OK = selftest()
        except Exception:   # noqa: BLE001 - any surprise means: keep the buffered pickle.dump
            _OK = False
    return _OK
